// Coarse-level context layers: the reference's LocalFeatureTransformer in its default coarse
// configuration - d_model 256, 8 heads, linear attention, no masks, any sequence of 'self' / 'cross' layers
// (network/module/transformer.py:34-57,78-96, attentions.py:19-46; called at network/net.py:74) on the token
// sets of both images, [N, L, 256] and [N, S, 256] float32.
//
// One encoder layer  x <- x + LN2(MLP([x, LN1(merge(attn(q(x), k(src), v(src))))]))  is three launches:
//
//   k_ctx_kv      one workgroup per 32 SOURCE tokens: k = Wk src, v = Wv src on the float32 matrix cores
//                 (v_mfma_f32_32x32x2_f32), K = elu(k) + 1, V = v / S, then this tile's share of the per-head
//                 KV[d][v] = sum_s K[s][d] V[s][v] (a 32 x 32 x 32 product per head, again on the matrix cores) and of
//                 Ksum[d] = sum_s K[s][d]; written as a partial, already in the operand-fragment order k_ctx_layer reads.
//   k_ctx_kv_sum  folds the partials of a sample in tile order (deterministic; no float atomics).
//   k_ctx_layer   one workgroup per 32 tokens of x, 4 waves, each owning a quarter of every layer's output
//                 channels: q projection -> elu + 1 -> per-head Q KV and the normaliser 1 / (Q . Ksum + eps) ->
//                 merge -> LayerNorm -> MLP (512 -> 512, ReLU, 512 -> 256) -> LayerNorm -> residual.  The activations
//                 of the tile never leave the CU: they sit in LDS as [k / 8][token][8] float32, which is both what
//                 an accumulator writes (4 consecutive channels per register quad: ds_write_b128) and what the next
//                 product's B operand reads (ds_read_b128 = 4 k-steps), without bank conflicts.  Weights stream from
//                 L2 straight into registers as A-operand fragments, packed so that a wave's 64 lanes read one
//                 contiguous 1 KiB per (32 output channels, 8 input channels); a four-chunk register ring keeps
//                 the loads ~2k cycles ahead of their use.
//
// Arithmetic is float32 throughout (float32 MFMA: 157 TFLOP/s dense peak on MI355X), so the result differs from the
// PyTorch module by summation order only.  Work: 2 * 655 360 flop per token and layer-call = 100.7 GFLOP per 640x480
// pair for the reference's 8 layers.
//
// k-index convention of every product here: chunk j covers k = 8j .. 8j+7; in step t (0..3) of the chunk the lanes
// of half h (= lane >> 5) contract k = 8j + 4h + t.  A-operand fragments (weights, KV) and B-operand reads
// (activations) both follow it, so a lane's 16 bytes are 4 consecutive k-steps.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "fm_internal.h"
#include "fmatch.h"

namespace {

using namespace fm;

constexpr int kD = 256;                  // d_model
constexpr int kH = 8, kHD = 32;          // heads x head dim
constexpr int kTok = 32;                 // tokens per workgroup
constexpr int kKvFloats = kH * kHD * kHD + kH * kHD;      // per sample: KV fragments [h][j][lane][4], then Ksum [h][d]

// packed layer: A-operand fragments of the six weight matrices, then the four LayerNorm vectors
constexpr size_t kOffQ = 0, kOffK = 65536, kOffV = 131072, kOffM = 196608, kOffW1 = 262144, kOffW2 = 524288,
                 kOffLn = 655360, kLayerFloats = kOffLn + 4 * kD;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Seg {
  const float* x;        // [N, L, 256] tokens this launch reads (k_ctx_kv: the SOURCE; k_ctx_layer: the image updated)
  float* out;            // k_ctx_layer: [N, L, 256] (may alias x)
  float* part;           // k_ctx_kv: [N * tiles][kKvFloats] partials
  float* kv;             // k_ctx_kv_sum: out; k_ctx_layer: in  [N][kKvFloats]
  int L;                 // tokens per sample
  int tiles;             // ceil(L / 32)
  float src_len;         // S of the attention: the source's token count (values / S ... * S)
};
struct TfArgs {
  Seg seg[2];
  int tiles0;            // workgroups of segment 0 (= N * seg[0].tiles); the rest belong to segment 1
  int N;
  const float* w;        // this layer's packed weights
#ifdef FM_DIAG_CTF
  float* diag;           // diagnostic build: [workgroups of k_ctx_layer][4 waves][16] shader-clock stamps
#endif
};

#ifdef FM_DIAG_CTF
#define CTF_STAMP(k) do { const long long c_ = __builtin_amdgcn_s_memtime(); if (lane == 0) dg[k] = (float)(c_ - c0_); } while (0)
#else
#define CTF_STAMP(k) do {} while (0)
#endif

// W [n_out][K] row-major -> fragments: ((rb * K/8 + j) * 64 + lane) * 4 + t  =  W[32 rb + (lane & 31)][8j + 4 (lane >> 5) + t]
__global__ void k_ctx_pack(const float* __restrict__ w, int n_out, int K, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_out * K) return;
  const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long c = i >> 8;
  const int j = (int)(c % (K / 8)), rb = (int)(c / (K / 8));
  dst[i] = w[(long)(32 * rb + (lane & 31)) * K + 8 * j + 4 * (lane >> 5) + t];
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.0f : __expf(x); }     // elu(x) + 1

// acc[i] += W[row block rb0 + i] . ACT  for a [K/8][32][8] float32 activation tile in LDS (KCH = K / 8 chunks).
// The weight fragments travel L2 -> registers through a ring of P chunks, the B operand (LDS) one chunk ahead.  Both
// are issued from inline asm with counted waits: written as plain loads, hipcc re-materialises every fragment right
// in front of its first use (the loads are from read-only memory), which exposes the full L2 latency per chunk
// (measured: 97-113 cycles per MFMA instead of 64).  VMEM returns in order, so `vmcnt(younger loads)` means the
// wanted chunk has landed; the stage drains its own loads before it returns.
#define CTF_GLOAD(dst, ptr, imm) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(dst) : "v"(ptr), "n"(imm) : "memory")
#define CTF_DSREAD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(imm) : "memory")

template <int VM>
__device__ __forceinline__ void ctf_wait(f32x4 (&w)[2], f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%3) lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(b) : "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void ctf_wait(f32x4 (&w)[4], f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(b) : "n"(VM) : "memory");
}

// one chunk: wait for its fragments, start the next B read, 4 k-steps x NB products, refill the ring slot
template <int NB, int P, int p, bool kTail>
__device__ __forceinline__ void ctf_step(f32x4 (&wa)[P][NB], f32x4 (&bq)[2], const char* (&wn)[NB], unsigned ba,
                                         f32x16 (&acc)[NB]) {
  ctf_wait<(kTail ? P - 1 - p : P - 1) * NB>(wa[p], bq[p & 1]);
  if (!kTail || p + 1 < P) CTF_DSREAD(bq[(p + 1) & 1], ba, (p + 1) * 1024);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[p][i][t], bq[p & 1][t], acc[i], 0, 0, 0);
  // refill the slot the products above have just read.  The registers of an in-flight load must never be touched by
  // compiler-generated code (it believes the asm's result is there at once): the ring is only ever named by the asm
  // statements and by products that follow a wait, and the tied "+v" keeps each slot in one physical register.
  if (!kTail) {
#pragma unroll
    for (int i = 0; i < NB; ++i) CTF_GLOAD(wa[p][i], wn[i], p * 1024 - 4096);
  }
}

template <int NB, int P, bool kTail>
__device__ __forceinline__ void ctf_group(f32x4 (&wa)[P][NB], f32x4 (&bq)[2], const char* (&wn)[NB], unsigned ba,
                                          f32x16 (&acc)[NB]) {
  ctf_step<NB, P, 0, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, 1, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, 2, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, 3, kTail>(wa, bq, wn, ba, acc);
  if constexpr (P == 8) {
    ctf_step<NB, P, 4, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, 5, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, 6, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, 7, kTail>(wa, bq, wn, ba, acc);
  }
}

template <int NB, int KCH>
__device__ __forceinline__ void gemm_stage(const float* __restrict__ wp, int rb0, const float* act, f32x16 (&acc)[NB],
                                           int lane) {
  constexpr int P = 16 / NB;     // chunks in flight: ~4k cycles of matrix-core work ahead of their use
  constexpr int G = KCH / P;
  static_assert(KCH % P == 0 && (P == 4 || P == 8) && G >= 2, "chunk count");
  // byte address of this lane's fragment of chunk P (the first one the loop prefetches), biased by +4096 so that the
  // chunk offsets p * 1024 - 4096 fit the signed 13-bit immediate
  const char* wn[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
    wn[i] = reinterpret_cast<const char*>(wp) + ((size_t)(rb0 + i) * KCH * 64 + lane) * 16 + 4096;
  unsigned ba = (unsigned)(uintptr_t)act + (unsigned)((lane & 31) * 2 + (lane >> 5)) * 16u;      // LDS byte address
  f32x4 wa[P][NB] = {}, bq[2] = {};
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int i = 0; i < NB; ++i) CTF_GLOAD(wa[p][i], wn[i], p * 1024 - 4096);
  CTF_DSREAD(bq[0], ba, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) wn[i] += P * 1024;
  for (int g = 0; g < G - 1; ++g) {      // every chunk this group prefetches exists
    ctf_group<NB, P, false>(wa, bq, wn, ba, acc);
#pragma unroll
    for (int i = 0; i < NB; ++i) wn[i] += P * 1024;
    ba += P * 1024;
  }
  ctf_group<NB, P, true>(wa, bq, wn, ba, acc);      // last P chunks: nothing left to prefetch, the waits count down to zero
}

// accumulator of output row block `rbg` (32 channels: register g <-> channel (g & 3) + 8 (g >> 2) + 4 h) -> activation
// tile: chunk 4 rbg + q, this lane's token, elements 4h .. 4h+3
template <class F>
__device__ __forceinline__ void store_act(float* buf, int rbg, const f32x16& acc, int lane, F f) {
  f32x4* p = reinterpret_cast<f32x4*>(buf) + (size_t)(4 * rbg) * 64 + (lane & 31) * 2 + (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = f(acc[4 * q + e], q, e);
    p[q * 64] = v;
  }
}

// tokens [tok0, tok0 + 32) of x [L, 256] -> [32 chunks][token][8]; rows beyond L are zero
__device__ __forceinline__ void load_tile(const float* __restrict__ x, int tok0, int L, float* buf, int tid) {
  // thread -> (token = tid & 31, chunk = tid >> 5 + 8 i): consecutive lanes write consecutive 32-byte LDS slots
  const int tok = tid & 31;
  const bool ok = tok0 + tok < L;
  const float* row = x + (size_t)(tok0 + tok) * kD;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (tid >> 5) + 8 * i;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (ok) {
      v0 = *reinterpret_cast<const f32x4*>(row + 8 * j);
      v1 = *reinterpret_cast<const f32x4*>(row + 8 * j + 4);
    }
    f32x4* p = reinterpret_cast<f32x4*>(buf) + ((size_t)j * 32 + tok) * 2;
    p[0] = v0;
    p[1] = v1;
  }
}

__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32); }

// ------------------------------------------------------------------------------------------------ k_ctx_kv
constexpr int kKtStride = 36;            // floats per channel row of the K / V transposes (32 tokens + pad, 16-byte aligned)
constexpr int kKvLdsFloats = 32 * 32 * 8 + 2 * kD * kKtStride;

__global__ __launch_bounds__(256) void k_ctx_kv(TfArgs a) {
  extern __shared__ float lds[];
  float* const xs = lds;                                  // [32][32][8] source tile
  float* const kt = lds + 32 * 32 * 8;                    // [256][36]  K, channel-major
  float* const vt = kt + kD * kKtStride;                  // [256][36]  V
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool s1 = (int)blockIdx.x >= a.tiles0;
  const Seg& sg = a.seg[s1 ? 1 : 0];
  const int wg = s1 ? (int)blockIdx.x - a.tiles0 : (int)blockIdx.x;
  const int b = wg / sg.tiles, tile = wg - b * sg.tiles, tok0 = tile * kTok;
  load_tile(sg.x + (size_t)b * sg.L * kD, tok0, sg.L, xs, tid);
  __syncthreads();
  const bool tok_ok = tok0 + r < sg.L;
  const float S = sg.src_len;
  // this wave: heads 2 wv, 2 wv + 1 = output row blocks 2 wv, 2 wv + 1 of both projections
  {
    f32x16 acc[2] = {};
    gemm_stage<2, 32>(a.w + kOffK, 2 * wv, xs, acc, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ch = 32 * (2 * wv + i) + (g & 3) + 8 * (g >> 2) + 4 * h;
        kt[ch * kKtStride + r] = tok_ok ? elu1(acc[i][g]) : 0.f;       // padding tokens contribute nothing
      }
  }
  {
    f32x16 acc[2] = {};
    gemm_stage<2, 32>(a.w + kOffV, 2 * wv, xs, acc, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ch = 32 * (2 * wv + i) + (g & 3) + 8 * (g >> 2) + 4 * h;
        vt[ch * kKtStride + r] = acc[i][g] / S;                          // values / S (attentions.py:41-42)
      }
  }
  __builtin_amdgcn_wave_barrier();        // the rows read below were written by this wave only
  __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
  float* const part = sg.part + (size_t)wg * kKvFloats;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hd = 2 * wv + i;
    // KV[d][v] = sum_tok K[tok][d] V[tok][v]: A = K^T (rows d), B = V (columns v); in step (c, t) half h contracts
    // token 16 h + 4 c + t
    const f32x4* ka = reinterpret_cast<const f32x4*>(kt + (hd * 32 + r) * kKtStride + 16 * h);
    const f32x4* vb = reinterpret_cast<const f32x4*>(vt + (hd * 32 + r) * kKtStride + 16 * h);
    f32x16 kv = {};
    float ks = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 ka4 = ka[c], vb4 = vb[c];
      ks += (ka4[0] + ka4[1]) + (ka4[2] + ka4[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) kv = __builtin_amdgcn_mfma_f32_32x32x2f32(ka4[t], vb4[t], kv, 0, 0, 0);
    }
    ks += other_half(ks);
    // register 4q + e of lane (v = r, h) is KV[d = 8q + 4h + e][v]: exactly lane's f32x4 of chunk q in the A-operand
    // fragment order of k_ctx_layer's Q KV product
    f32x4* o = reinterpret_cast<f32x4*>(part) + (size_t)(hd * 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v4 = {kv[4 * q], kv[4 * q + 1], kv[4 * q + 2], kv[4 * q + 3]};
      o[q * 64] = v4;
    }
    if (h == 0) part[kH * kHD * kHD + hd * 32 + r] = ks;
  }
}

// kv[b][o] = sum over the sample's tiles, in tile order
__global__ __launch_bounds__(256) void k_ctx_kv_sum(TfArgs a) {
  const Seg& sg = a.seg[blockIdx.z];
  const int o = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (o >= kKvFloats) return;
  const float* p = sg.part + (size_t)b * sg.tiles * kKvFloats + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = 0;
  for (; t + 4 <= sg.tiles; t += 4) {
    s0 += p[(size_t)t * kKvFloats];
    s1 += p[(size_t)(t + 1) * kKvFloats];
    s2 += p[(size_t)(t + 2) * kKvFloats];
    s3 += p[(size_t)(t + 3) * kKvFloats];
  }
  for (; t < sg.tiles; ++t) s0 += p[(size_t)t * kKvFloats];
  sg.kv[(size_t)b * kKvFloats + o] = (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------------------------------------ k_ctx_layer
constexpr int kLayerLdsFloats = 2 * 64 * 32 * 8 + 2 * 4 * 32;      // x | msg, hidden (its first half: Q / attention), 2 reductions

// LayerNorm over the 256 channels of every token: this wave holds 64 of them (2 row blocks x 16 registers x 2 halves)
__device__ __forceinline__ void layer_norm(f32x16 (&y)[2], const float* __restrict__ gamma, const float* __restrict__ beta,
                                           float* red0, float* red1, int wv, int lane) {
  const int r = lane & 31, h = lane >> 5;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) s += y[i][g];
  s += other_half(s);
  if (h == 0) red0[wv * 32 + r] = s;
  __syncthreads();
  const float mean = ((red0[r] + red0[32 + r]) + (red0[64 + r] + red0[96 + r])) * (1.0f / kD);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      y[i][g] -= mean;
      v = __builtin_fmaf(y[i][g], y[i][g], v);
    }
  v += other_half(v);
  if (h == 0) red1[wv * 32 + r] = v;
  __syncthreads();
  const float var = ((red1[r] + red1[32 + r]) + (red1[64 + r] + red1[96 + r])) * (1.0f / kD);
  const float rstd = 1.0f / __builtin_sqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * (2 * wv + i) + 8 * q + 4 * h;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + ch), b4 = *reinterpret_cast<const f32x4*>(beta + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[i][4 * q + e] = __builtin_fmaf(y[i][4 * q + e] * rstd, g4[e], b4[e]);
    }
}

__global__ __launch_bounds__(256) void k_ctx_layer(TfArgs a) {
  extern __shared__ float lds[];
  float* const xm = lds;                          // chunks 0..31: x, 32..63: LN1(merge(attention))
  float* const hb = lds + 64 * 32 * 8;            // chunks 0..63: MLP hidden; before that chunks 0..31: Q, then the attention output
  float* const red0 = hb + 64 * 32 * 8;
  float* const red1 = red0 + 4 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool s1 = (int)blockIdx.x >= a.tiles0;
  const Seg& sg = a.seg[s1 ? 1 : 0];
  const int wg = s1 ? (int)blockIdx.x - a.tiles0 : (int)blockIdx.x;
  const int b = wg / sg.tiles, tile = wg - b * sg.tiles, tok0 = tile * kTok;
#ifdef FM_DIAG_CTF
  const long long c0_ = __builtin_amdgcn_s_memtime();
  float* const dg = a.diag + ((size_t)blockIdx.x * 4 + wv) * 16;
#endif
  load_tile(sg.x + (size_t)b * sg.L * kD, tok0, sg.L, xm, tid);
  __syncthreads();
  CTF_STAMP(0);
  const float* const kvp = sg.kv + (size_t)b * kKvFloats;
  const float* const ln = a.w + kOffLn;

  // ---- Q = elu(Wq x) + 1: this wave's two heads ----
  {
    f32x16 acc[2] = {};
    gemm_stage<2, 32>(a.w + kOffQ, 2 * wv, xm, acc, lane);
    CTF_STAMP(1);
#pragma unroll
    for (int i = 0; i < 2; ++i) store_act(hb, 2 * wv + i, acc[i], lane, [](float v, int, int) { return elu1(v); });
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // ---- per head: (Q KV) / (Q . Ksum + eps) * S, written over Q (attentions.py:43-46) ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hd = 2 * wv + i;
    const f32x4* qa = reinterpret_cast<const f32x4*>(hb) + (size_t)(4 * hd) * 64 + r * 2 + h;
    const f32x4* ka = reinterpret_cast<const f32x4*>(kvp) + (size_t)(4 * hd) * 64 + lane;
    f32x16 o = {};
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 qb = qa[j * 64], kf = ka[j * 64];
      const f32x4 ks = *reinterpret_cast<const f32x4*>(kvp + kH * kHD * kHD + hd * 32 + 8 * j + 4 * h);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        den = __builtin_fmaf(qb[t], ks[t], den);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qb[t], o, 0, 0, 0);
      }
    }
    den += other_half(den);
    const float z = 1.0f / (den + 1e-6f);
    const float S = sg.src_len;
    store_act(hb, hd, o, lane, [z, S](float v, int, int) { return v * z * S; });
  }
  CTF_STAMP(2);
  __syncthreads();
  CTF_STAMP(3);
  // ---- merge + LayerNorm 1 -> chunks 32..63 of the MLP input ----
  {
    f32x16 acc[2] = {};
    gemm_stage<2, 32>(a.w + kOffM, 2 * wv, hb, acc, lane);
    CTF_STAMP(4);
    layer_norm(acc, ln, ln + kD, red0, red1, wv, lane);       // (its barriers also fence the reads of hb above)
#pragma unroll
    for (int i = 0; i < 2; ++i) store_act(xm, 8 + 2 * wv + i, acc[i], lane, [](float v, int, int) { return v; });
  }
  __syncthreads();
  CTF_STAMP(5);
  // ---- MLP: hidden = relu(W1 [x, msg]) ----
  {
    f32x16 acc[4] = {};
    gemm_stage<4, 64>(a.w + kOffW1, 4 * wv, xm, acc, lane);
    CTF_STAMP(6);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_act(hb, 4 * wv + i, acc[i], lane, [](float v, int, int) { return fmaxf(v, 0.f); });
  }
  __syncthreads();
  CTF_STAMP(7);
  // ---- W2 hidden -> LayerNorm 2 -> residual ----
  {
    f32x16 acc[2] = {};
    gemm_stage<2, 64>(a.w + kOffW2, 2 * wv, hb, acc, lane);
    CTF_STAMP(8);
    layer_norm(acc, ln + 2 * kD, ln + 3 * kD, red0, red1, wv, lane);
    CTF_STAMP(9);
    if (tok0 + r < sg.L) {
      float* const orow = sg.out + ((size_t)b * sg.L + tok0 + r) * kD;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rbg = 2 * wv + i;
        const f32x4* xp = reinterpret_cast<const f32x4*>(xm) + (size_t)(4 * rbg) * 64 + r * 2 + h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 xv = xp[q * 64];
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = xv[e] + acc[i][4 * q + e];
          *reinterpret_cast<f32x4*>(orow + 32 * rbg + 8 * q + 4 * h) = ov;
        }
      }
    }
  }
  CTF_STAMP(10);
}

constexpr int kMaxLayers = 32;

size_t ws_floats(int N, int L, int S) {
  const size_t t0 = (size_t)N * ((L + kTok - 1) / kTok), t1 = (size_t)N * ((S + kTok - 1) / kTok);
  size_t n = (t0 + t1) * kKvFloats + 2 * (size_t)N * kKvFloats;
#ifdef FM_DIAG_CTF
  n += (t0 + t1) * 4 * 16;
#endif
  return n;
}

}  // namespace

extern "C" size_t fm_coarse_tf_packed_bytes(int n_layers) {
  return n_layers > 0 && n_layers <= kMaxLayers ? (size_t)n_layers * kLayerFloats * 4 : 0;
}

extern "C" int fm_coarse_tf_workspace_bytes(int N, int L, int S, size_t* bytes) {
  if (!bytes) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  *bytes = ws_floats(N, L, S) * 4;
  return FM_OK;
}

// layers[l]: 10 device pointers in state-dict order - q_proj, k_proj, v_proj, merge .weight [256,256]; mlp.0.weight
// [512,512]; mlp.2.weight [256,512]; norm1.weight, norm1.bias, norm2.weight, norm2.bias [256]
extern "C" int fm_coarse_tf_pack_weights(const float* const* const* layers, int n_layers, void* packed, void* stream) {
  if (!layers || !packed) return FM_E_NULL;
  if (n_layers <= 0 || n_layers > kMaxLayers) return FM_E_UNSUPPORTED;
  for (int l = 0; l < n_layers; ++l) {
    if (!layers[l]) return FM_E_NULL;
    for (int i = 0; i < 10; ++i)
      if (!layers[l][i]) return FM_E_NULL;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t off[6] = {kOffQ, kOffK, kOffV, kOffM, kOffW1, kOffW2};
  const int n_out[6] = {256, 256, 256, 256, 512, 256}, n_in[6] = {256, 256, 256, 256, 512, 512};
  for (int l = 0; l < n_layers; ++l) {
    float* dst = (float*)packed + (size_t)l * kLayerFloats;
    for (int i = 0; i < 6; ++i) {
      const int n = n_out[i] * n_in[i];
      hipLaunchKernelGGL(k_ctx_pack, dim3((n + 255) / 256), dim3(256), 0, st, layers[l][i], n_out[i], n_in[i], dst + off[i]);
    }
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipMemcpyAsync(dst + kOffLn + (size_t)i * kD, layers[l][6 + i], kD * 4, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  return (int)hipGetLastError();
}

// layer_kinds[l]: 0 = 'self', 1 = 'cross' (transformer.py:88-95).  out0 / out1 [N,L,256] / [N,S,256] must not alias
// the inputs (the first layer reads feat*, every later one updates out* in place).
extern "C" int fm_coarse_transformer(const float* feat0, const float* feat1, int N, int L, int S, int C, int nhead,
                                     const int* layer_kinds, int n_layers, const void* packed, void* workspace,
                                     size_t workspace_bytes, float* out0, float* out1, void* stream) {
  if (!feat0 || !feat1 || !layer_kinds || !packed || !workspace || !out0 || !out1) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  if (C != kD || nhead != kH || n_layers <= 0 || n_layers > kMaxLayers) return FM_E_UNSUPPORTED;
  for (int l = 0; l < n_layers; ++l)
    if (layer_kinds[l] != 0 && layer_kinds[l] != 1) return FM_E_UNSUPPORTED;
  if (feat0 == out0 || feat1 == out1 || out0 == out1) return FM_E_UNSUPPORTED;
  if (workspace_bytes < ws_floats(N, L, S) * 4 || ((uintptr_t)workspace & 15)) return FM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long set_kv = 0, set_layer = 0;
  hipError_t e = ensure_dynamic_lds(&k_ctx_kv, kKvLdsFloats * 4, &set_kv);
  if (e != hipSuccess) return (int)e;
  e = ensure_dynamic_lds(&k_ctx_layer, kLayerLdsFloats * 4, &set_layer);
  if (e != hipSuccess) return (int)e;

  const int tl[2] = {(L + kTok - 1) / kTok, (S + kTok - 1) / kTok}, len[2] = {L, S};
  float* part[2];
  float* kv[2];
  part[0] = (float*)workspace;
  part[1] = part[0] + (size_t)N * tl[0] * kKvFloats;
  kv[0] = part[1] + (size_t)N * tl[1] * kKvFloats;
  kv[1] = kv[0] + (size_t)N * kKvFloats;
#ifdef FM_DIAG_CTF
  float* const diag = kv[1] + (size_t)N * kKvFloats;
#endif
  const float* cur[2] = {feat0, feat1};
  float* out[2] = {out0, out1};

  // one encoder layer on `nseg` (image, source) pairs: x[img[s]] <- layer(x[img[s]], x[src[s]])
  auto run = [&](const float* w, int nseg, const int* img, const int* src) -> hipError_t {
    TfArgs a{};
    a.N = N;
    a.w = w;
#ifdef FM_DIAG_CTF
    a.diag = diag;
#endif
    // K / V side: the SOURCE tokens
    int tiles_kv = 0;
    for (int s = 0; s < nseg; ++s) {
      Seg& g = a.seg[s];
      g.x = cur[src[s]]; g.out = nullptr; g.part = part[src[s]]; g.kv = kv[src[s]];
      g.L = len[src[s]]; g.tiles = tl[src[s]]; g.src_len = (float)len[src[s]];
      tiles_kv += N * g.tiles;
    }
    a.tiles0 = N * a.seg[0].tiles;
    hipLaunchKernelGGL(k_ctx_kv, dim3(tiles_kv), dim3(256), kKvLdsFloats * 4, st, a);
    hipLaunchKernelGGL(k_ctx_kv_sum, dim3((kKvFloats + 255) / 256, N, nseg), dim3(256), 0, st, a);
    // query side: the tokens that are updated
    int tiles_x = 0;
    for (int s = 0; s < nseg; ++s) {
      Seg& g = a.seg[s];
      g.x = cur[img[s]]; g.out = out[img[s]]; g.part = nullptr; g.kv = kv[src[s]];
      g.L = len[img[s]]; g.tiles = tl[img[s]]; g.src_len = (float)len[src[s]];
      tiles_x += N * g.tiles;
    }
    a.tiles0 = N * a.seg[0].tiles;
    hipLaunchKernelGGL(k_ctx_layer, dim3(tiles_x), dim3(256), kLayerLdsFloats * 4, st, a);
    for (int s = 0; s < nseg; ++s) cur[img[s]] = out[img[s]];
    return hipGetLastError();
  };

  for (int l = 0; l < n_layers; ++l) {
    const float* w = (const float*)packed + (size_t)l * kLayerFloats;
    if (layer_kinds[l] == 0) {
      const int img[2] = {0, 1}, src[2] = {0, 1};
      e = run(w, 2, img, src);
      if (e != hipSuccess) return (int)e;
    } else {
      const int i0[1] = {0}, s0[1] = {1};
      e = run(w, 1, i0, s0);                    // feat0 <- layer(feat0, feat1)
      if (e != hipSuccess) return (int)e;
      const int i1[1] = {1}, s1[1] = {0};
      e = run(w, 1, i1, s1);                    // feat1 <- layer(feat1, UPDATED feat0)  (transformer.py:93-94)
      if (e != hipSuccess) return (int)e;
    }
  }
  return FM_OK;
}
