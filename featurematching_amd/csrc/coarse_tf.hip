// Coarse-level context layers: the reference's LocalFeatureTransformer in its default coarse
// configuration - d_model 256, 8 heads, linear attention, no masks, any sequence of 'self' / 'cross' layers
// (network/module/transformer.py:34-57,78-96, attentions.py:19-46; called at network/net.py:74) on the token
// sets of both images, [N, L, 256] and [N, S, 256] float32.
//
// One encoder layer  x <- x + LN2(MLP([x, LN1(merge(attn(q(x), k(src), v(src))))]))  is three launches:
//
//   k_ctx_kv      one workgroup per 32 SOURCE tokens: k = Wk src, v = Wv src, K = elu(k) + 1, V = v / S, then this
//                 tile's share of the per-head KV[d][v] = sum_s K[s][d] V[s][v] (a 32 x 32 x 32 product per head on
//                 the float32 matrix cores) and of Ksum[d] = sum_s K[s][d]; written as a partial, already in the
//                 operand-fragment order k_ctx_layer reads.
//   k_ctx_kv_sum  folds the partials of a sample in tile order (deterministic; no float atomics).
//   k_ctx_layer   one workgroup per 32 tokens of x, 4 waves, each owning a quarter of every layer's output
//                 channels: q projection -> elu + 1 -> per-head Q KV and the normaliser 1 / (Q . Ksum + eps) ->
//                 merge -> LayerNorm -> MLP (512 -> 512, ReLU, 512 -> 256) -> LayerNorm -> residual.  The activations
//                 of the tile never leave the CU.
//
// The six linear layers (655 360 MACs per token, >98 % of the work) are float32-EQUIVALENT products on the float16
// matrix cores: x = hi + lo with hi = f16(2^e x), lo = f16(2^e x - hi) (22 significant bits), W likewise, and
// W x ~ hi_W hi_x + lo_W hi_x + hi_W lo_x = 3 x v_mfma_f32_32x32x16_f16 per 16 k - 5.3x the rate of the float32
// MFMA (8 x 64 cycles against 3 x 32).  Every operand carries an exact power-of-two scale that brings its largest
// magnitude to [2^13, 2^14): the matrix cores flush float16 SUBNORMAL inputs, so an unscaled lo half would vanish
// for every |x| < 2^-3.  Weights: one scale per matrix, fixed at pack time.  Activations: one scale per TOKEN, from
// the token's exact maximum over all channels (one LDS reduction per operand).  The accumulator is scaled back in
// the epilogue (exact).  The per-head 32 x 32 attention products stay on the float32 MFMA.
//
// Operand traffic: activations sit in LDS as float16 planes [k / 16][k % 16 / 8][token][8] (hi) + the same for lo -
// what the next product's B operand reads with one ds_read_b128 per plane and k-step; an accumulator register quad is
// 4 consecutive channels = one ds_write_b64 per plane.  Weights stream from L2 straight into registers as A-operand
// fragments (1 KiB hi + 1 KiB lo per 32 output channels and 16 input channels, contiguous) through a register ring
// of 16 / NB k-steps, issued from inline asm with counted waits: written as plain loads, hipcc re-materialises every
// fragment right in front of its first use (the loads are from read-only memory), which exposes the full L2
// latency per k-step.  At 2.1 MB of fragments per tile and layer the kernel is bound by the L2 -> CU path
// (64 B / clk / CU), not by the matrix cores.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "fm_internal.h"
#include "fmatch.h"

namespace {

using namespace fm;

constexpr int kD = 256;                  // d_model
constexpr int kH = 8, kHD = 32;          // heads x head dim
constexpr int kTok = 32;                 // tokens per workgroup
constexpr int kKvFloats = kH * kHD * kHD + kH * kHD;      // per sample: KV fragments [h][j][lane][4], then Ksum [h][d]
static_assert(kKvFloats % 32 == 0, "k_ctx_kv_sum: 32 outputs per workgroup");

// packed layer (bytes): float16 A-operand fragments of the six weight matrices (hi + lo = 4 bytes per weight), then
// a float header: the four LayerNorm vectors, 1 / scale of every matrix, the bound of |LN1 output|, the matrices'
// largest magnitudes (bit patterns; scratch of the packing)
constexpr size_t kFragQ = 0, kFragK = 65536 * 4, kFragV = 2 * 65536 * 4, kFragM = 3 * 65536 * 4, kFragW1 = 4 * 65536 * 4,
                 kFragW2 = kFragW1 + 262144 * 4, kFragEnd = kFragW2 + 131072 * 4;
constexpr int kHdrLn = 0, kHdrWinv = 4 * kD, kHdrMsgBound = kHdrWinv + 6, kHdrAbsmax = kHdrMsgBound + 1, kHdrFloats = kHdrAbsmax + 6 + 3;
constexpr size_t kLayerBytes = kFragEnd + (size_t)kHdrFloats * 4;
static_assert(kLayerBytes % 16 == 0, "layer stride keeps the fragments 16-byte aligned");

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct Seg {
  const float* x;        // [N, L, 256] tokens this launch reads (k_ctx_kv: the SOURCE; k_ctx_layer: the image updated)
  float* out;            // k_ctx_layer: [N, L, 256] (may alias x)
  float* part;           // k_ctx_kv: [N * tiles][kKvFloats] partials
  float* kv;             // k_ctx_kv_sum: out; k_ctx_layer: in  [N][kKvFloats]
  int L;                 // tokens per sample
  int tiles;             // ceil(L / 32)
  float src_len;         // S of the attention: the source's token count (values / S ... * S)
  const unsigned char* mask;   // optional padding mask [N, L] of the tokens this launch reads (1 = a real token; attentions.py:35-40)
};
struct TfArgs {
  Seg seg[2];
  int tiles0;            // workgroups of segment 0 (= N * seg[0].tiles); the rest belong to segment 1
  int N;
  const char* w;         // this layer's packed weights
#ifdef FM_DIAG_CTF
  float* diag;           // diagnostic build: [workgroups of k_ctx_layer][4 waves][16] shader-clock stamps
#endif
};

#ifdef FM_DIAG_CTF
#define CTF_STAMP(k) do { const long long c_ = __builtin_amdgcn_s_memtime(); if (lane == 0) dg[k] = (float)(c_ - c0_); } while (0)
#else
#define CTF_STAMP(k) do {} while (0)
#endif

// exact power of two s with amax * s in [2^13, 2^14) and its inverse (zero, denormal or non-finite amax: 1)
__host__ __device__ __forceinline__ void pow2_scale(float amax, float& s, float& inv) {
  unsigned bits;
  __builtin_memcpy(&bits, &amax, 4);
  const int e = (int)((bits >> 23) & 0xffu);           // amax in [2^(e-127), 2^(e-126))
  int k = 140 - e;                                     // 13 - (e - 127)
  if (e == 0 || e == 255) k = 0;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const unsigned sb = (unsigned)(127 + k) << 23, ib = (unsigned)(127 - k) << 23;
  __builtin_memcpy(&s, &sb, 4);
  __builtin_memcpy(&inv, &ib, 4);
}

// ------------------------------------------------------------------------------------------------ packing
__global__ void k_ctx_absmax(const float* __restrict__ w, int n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

// W [n_out][K] row-major -> per (32-row block rb, 16-k step c): 64 lanes x 8 float16 of hi (1 KiB), then of lo;
// lane (r, h) holds W[32 rb + r][16 c + 8 h .. + 7] * scale
__global__ void k_ctx_pack16(const float* __restrict__ w, int n_out, int K, const unsigned* __restrict__ absmax_bits,
                             half8* __restrict__ dst, float* __restrict__ winv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (rb, c, lane)
  if (i >= n_out * K / 8) return;
  float s, inv;
  pow2_scale(__uint_as_float(*absmax_bits), s, inv);
  if (i == 0) *winv = inv;
  const int lane = i & 63, c = (i >> 6) % (K / 16), rb = (i >> 6) / (K / 16);
  const float* row = w + (size_t)(32 * rb + (lane & 31)) * K + 16 * c + 8 * (lane >> 5);
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = row[e] * s;
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  half8* o = dst + (size_t)(i >> 6) * 128 + lane;
  o[0] = hi;
  o[64] = lo;
}

// LayerNorm vectors into the header + the bound |LN1(.)| <= max|gamma| sqrt(d - 1) + max|beta| of the merged message
__global__ void k_ctx_pack_ln(const float* g1, const float* b1, const float* g2, const float* b2, float* hdr) {
  const int t = threadIdx.x;       // 256 threads
  hdr[kHdrLn + t] = g1[t];
  hdr[kHdrLn + kD + t] = b1[t];
  hdr[kHdrLn + 2 * kD + t] = g2[t];
  hdr[kHdrLn + 3 * kD + t] = b2[t];
  __shared__ float sg[4], sb[4];
  float mg = fabsf(g1[t]), mb = fabsf(b1[t]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { mg = fmaxf(mg, __shfl_xor(mg, o)); mb = fmaxf(mb, __shfl_xor(mb, o)); }
  if ((t & 63) == 0) { sg[t >> 6] = mg; sb[t >> 6] = mb; }
  __syncthreads();
  if (t == 0)
    hdr[kHdrMsgBound] = fmaxf(fmaxf(sg[0], sg[1]), fmaxf(sg[2], sg[3])) * 16.0f + fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.0f : __expf(x); }     // elu(x) + 1
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32); }

// ------------------------------------------------------------------------------------------------ the split product
#define CTF_GLOAD(dst, ptr, imm) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(dst) : "v"(ptr), "n"(imm) : "memory")
#define CTF_DSREAD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(imm) : "memory")

template <int VM>
__device__ __forceinline__ void ctf_wait(half8 (&w)[2][2], half8 (&b)[2]) {
  asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)"
               : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(b[0]), "+v"(b[1]) : "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void ctf_wait(half8 (&w)[1][2], half8 (&b)[2]) {
  asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(b[0]), "+v"(b[1]) : "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void ctf_wait(half8 (&w)[4][2], half8 (&b)[2]) {
  asm volatile("s_waitcnt vmcnt(%10) lgkmcnt(0)"
               : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]),
                 "+v"(w[3][1]), "+v"(b[0]), "+v"(b[1]) : "n"(VM) : "memory");
}

// one 16-k step: wait for its fragments, start the next B reads, 3 x NB products, refill the ring slot.
// The registers of an in-flight load must never be touched by compiler-generated code (it believes the asm's result
// is there at once): the ring is only ever named by the asm statements and by products that follow a wait, and the
// tied "+v" keeps each slot in one physical register.
template <int NB, int P, int LO, int p, bool kTail>
__device__ __forceinline__ void ctf_step(half8 (&wa)[P][NB][2], half8 (&bq)[2][2], const char* (&wn)[NB], unsigned ba,
                                         f32x16 (&acc)[NB]) {
  ctf_wait<(kTail ? P - 1 - p : P - 1) * 2 * NB>(wa[p], bq[p & 1]);
  if (!kTail || p + 1 < P) {
    CTF_DSREAD(bq[(p + 1) & 1][0], ba, (p + 1) * 1024);
    CTF_DSREAD(bq[(p + 1) & 1][1], ba, (p + 1) * 1024 + LO);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[p][i][0], bq[p & 1][0], acc[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[p][i][1], bq[p & 1][0], acc[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[p][i][0], bq[p & 1][1], acc[i], 0, 0, 0);
  if (!kTail) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      CTF_GLOAD(wa[p][i][0], wn[i], 0);
      CTF_GLOAD(wa[p][i][1], wn[i], 1024);
      wn[i] += 2048;
    }
  }
}

template <int NB, int P, int LO, bool kTail>
__device__ __forceinline__ void ctf_group(half8 (&wa)[P][NB][2], half8 (&bq)[2][2], const char* (&wn)[NB], unsigned ba,
                                          f32x16 (&acc)[NB]) {
  ctf_step<NB, P, LO, 0, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, LO, 1, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, LO, 2, kTail>(wa, bq, wn, ba, acc);
  ctf_step<NB, P, LO, 3, kTail>(wa, bq, wn, ba, acc);
  if constexpr (P == 8) {
    ctf_step<NB, P, LO, 4, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, LO, 5, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, LO, 6, kTail>(wa, bq, wn, ba, acc);
    ctf_step<NB, P, LO, 7, kTail>(wa, bq, wn, ba, acc);
  }
}

// acc[i] += W[row block rb0 + i] . ACT over KS k-steps of 16; `frag` = the matrix's fragments, `hi` = LDS address of the
// hi plane's first k-step (the lo plane LO bytes behind it)
// (RB_HI: accumulators NB/2 .. NB-1 take row blocks rb0 + RB_HI + ..: two matrices of one fragment region in one pass)
template <int NB, int KS, int LO, int RB_HI = 0>
__device__ __forceinline__ void gemm_stage(const char* __restrict__ frag, int rb0, const void* hi, f32x16 (&acc)[NB], int lane) {
  constexpr int P = NB == 1 ? 8 : 16 / NB;     // k-steps in flight (NB = 1: 8 x 2 KiB per wave, two waves per SIMD)
  constexpr int G = KS / P;
  static_assert(KS % P == 0 && (P == 4 || P == 8) && G >= 2, "step count");
  const char* wn[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int rb = RB_HI ? rb0 + (i % (NB / 2)) + RB_HI * (i / (NB / 2)) : rb0 + i;
    wn[i] = frag + ((size_t)rb * KS * 128 + lane) * 16;
  }
  unsigned ba = (unsigned)(uintptr_t)hi + (unsigned)lane * 16u;      // this lane's (token, k half) slot of a k-step
  half8 wa[P][NB][2] = {}, bq[2][2] = {};
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      CTF_GLOAD(wa[p][i][0], wn[i], 0);
      CTF_GLOAD(wa[p][i][1], wn[i], 1024);
      wn[i] += 2048;
    }
  CTF_DSREAD(bq[0][0], ba, 0);
  CTF_DSREAD(bq[0][1], ba, LO);
  for (int g = 0; g < G - 1; ++g) {      // every k-step this group prefetches exists
    ctf_group<NB, P, LO, false>(wa, bq, wn, ba, acc);
    ba += P * 1024;
  }
  ctf_group<NB, P, LO, true>(wa, bq, wn, ba, acc);      // last P steps: nothing left to prefetch, the waits count down
}

// accumulator values v[g] (register g <-> channel (g & 3) + 8 (g >> 2) + 4 h of output row block `rbg`), already
// multiplied by the token's scale -> the planes: k-step 2 rbg + (q >> 1), k half q & 1, this token, elements 4h .. 4h+3
__device__ __forceinline__ void store_planes(char* hi, int lo_off, int rbg, const f32x16& v, int lane) {
  char* p = hi + (size_t)(2 * rbg) * 1024 + (lane & 31) * 16 + (lane >> 5) * 8;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    half4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h4[e] = (_Float16)v[4 * q + e];
      l4[e] = (_Float16)(v[4 * q + e] - (float)h4[e]);
    }
    char* o = p + (q >> 1) * 1024 + (q & 1) * 512;
    *reinterpret_cast<half4*>(o) = h4;
    *reinterpret_cast<half4*>(o + lo_off) = l4;
  }
}

// tokens [tok0, tok0 + 32) of x [L, 256] -> planes (k-steps 0..15) scaled per token by the power of two that fits
// max(|x| of the token, floor_max); scale and 1 / scale per token -> xsc[32], xinv[32].  Rows beyond L are zero.  Contains two barriers.
template <int NG = 8>        // NG = threads / 32: 8-channel groups handled side by side
__device__ __forceinline__ void load_tile(const float* __restrict__ x, int tok0, int L, char* hi, int lo_off, float* red8,
                                          float* xinv, float* xsc, float floor_max, int tid) {
  // thread -> (token = tid & 31, 8-channel groups (tid >> 5) + NG i)
  constexpr int NI = 32 / NG;
  const int tok = tid & 31, g = tid >> 5;
  const bool ok = tok0 + tok < L;
  const float* row = x + (size_t)(tok0 + tok) * kD;
  f32x4 v[NI][2];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = g + NG * i;
    v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
      v[i][0] = *reinterpret_cast<const f32x4*>(row + 8 * j);
      v[i][1] = *reinterpret_cast<const f32x4*>(row + 8 * j + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(v[i][0][e]), fabsf(v[i][1][e])));
  }
  red8[g * 32 + tok] = m;
  __syncthreads();
  m = floor_max;
#pragma unroll
  for (int k = 0; k < NG; ++k) m = fmaxf(m, red8[k * 32 + tok]);
  float s, inv;
  pow2_scale(m, s, inv);
  if (g == 0) { xinv[tok] = inv; xsc[tok] = s; }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = g + NG * i;                 // channels 8j .. 8j+7 = k-step j >> 1, half j & 1: one 16-byte slot
    half8 h8, l8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xs = (e < 4 ? v[i][0][e] : v[i][1][e - 4]) * s;
      h8[e] = (_Float16)xs;
      l8[e] = (_Float16)(xs - (float)h8[e]);
    }
    char* o = hi + (size_t)(j >> 1) * 1024 + (j & 1) * 512 + tok * 16;
    *reinterpret_cast<half8*>(o) = h8;
    *reinterpret_cast<half8*>(o + lo_off) = l8;
  }
  __syncthreads();
}

// largest |v| of every token over all 256 / 512 channels of the workgroup (this wave holds NB x 16 x 2 of them);
// contains one barrier.  Returns the token's power-of-two scale and its inverse.
template <int NB, int NW = 4>
__device__ __forceinline__ void token_scale(const f32x16 (&v)[NB], float* redm, int wv, int lane, float& s, float& inv) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) m = fmaxf(m, fabsf(v[i][g]));
  m = fmaxf(m, other_half(m));
  const int r = lane & 31;
  if (lane < 32) redm[wv * 32 + r] = m;
  __syncthreads();
  m = fmaxf(fmaxf(redm[r], redm[32 + r]), fmaxf(redm[64 + r], redm[96 + r]));
  if constexpr (NW == 8) m = fmaxf(m, fmaxf(fmaxf(redm[128 + r], redm[160 + r]), fmaxf(redm[192 + r], redm[224 + r])));
  pow2_scale(m, s, inv);
}

// ------------------------------------------------------------------------------------------------ k_ctx_kv
constexpr int kKtStride = 36;            // floats per channel row of the K / V transposes (32 tokens + pad, 16-byte aligned)
constexpr int kPlane = 16 * 1024;        // bytes of one float16 plane of 256 channels x 32 tokens
// the K / V transposes reuse the bytes of the source planes (dead once both projections are done): 75 KB per workgroup,
// two workgroups per CU - the 300 tiles of a self layer (both images in one launch) then run in ONE round
constexpr int kKvLdsBytes = 2 * kD * kKtStride * 4 + (8 * 32 + 2 * 32) * 4;
static_assert(2 * kD * kKtStride * 4 >= 2 * kPlane, "the planes fit under the transposes");

__global__ __launch_bounds__(256) void k_ctx_kv(TfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const xh = lds;                                                   // source tile: hi plane, lo plane
  float* const kt = reinterpret_cast<float*>(lds);                        // [256][36]  K, channel-major (over the planes)
  float* const vt = kt + kD * kKtStride;                                  // [256][36]  V
  float* const red8 = vt + kD * kKtStride;
  float* const xinv = red8 + 8 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool s1 = (int)blockIdx.x >= a.tiles0;
  const Seg& sg = a.seg[s1 ? 1 : 0];
  const int wg = s1 ? (int)blockIdx.x - a.tiles0 : (int)blockIdx.x;
  const int b = wg / sg.tiles, tile = wg - b * sg.tiles, tok0 = tile * kTok;
  load_tile(sg.x + (size_t)b * sg.L * kD, tok0, sg.L, xh, kPlane, red8, xinv, xinv + 32, 0.f, tid);
  const float* const hdr = reinterpret_cast<const float*>(a.w + kFragEnd);
  // padded source tokens (kv_mask, attentions.py:38-40): K = 0 removes their K V^T term and their share of Ksum; their
  // V needs no mask of its own then
  const bool tok_ok = tok0 + r < sg.L && (!sg.mask || sg.mask[(size_t)b * sg.L + tok0 + r] != 0);
  const float S = sg.src_len, xi = xinv[r];
  // this wave: heads 2 wv, 2 wv + 1 = output row blocks 2 wv, 2 wv + 1 of both projections, in ONE pass over the source
  // planes (the fragments of Wv follow those of Wk: row blocks 8 .. 15 of the same region)
  {
    f32x16 acc[4] = {};
    gemm_stage<4, 16, kPlane, 8>(a.w + kFragK, 2 * wv, xh, acc, lane);
    __syncthreads();                      // every wave has read the planes: their bytes become the transposes
    const float fk = hdr[kHdrWinv + 1] * xi, fv = hdr[kHdrWinv + 2] * xi;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ch = 32 * (2 * wv + i) + (g & 3) + 8 * (g >> 2) + 4 * h;
        kt[ch * kKtStride + r] = tok_ok ? elu1(acc[i][g] * fk) : 0.f;       // padding tokens contribute nothing
        vt[ch * kKtStride + r] = acc[2 + i][g] * fv / S;                     // values / S (attentions.py:41-42)
      }
  }
  __builtin_amdgcn_wave_barrier();        // the rows read below were written by this wave only
  __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
  float* const part = sg.part + (size_t)wg * kKvFloats;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hd = 2 * wv + i;
    // KV[d][v] = sum_tok K[tok][d] V[tok][v] on the float32 MFMA: A = K^T (rows d), B = V (columns v); in step (c, t)
    // half h contracts token 16 h + 4 c + t
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
    // register 4q + e of lane (v = r, h) is KV[d = 8q + 4h + e][v]: exactly the lane's f32x4 of chunk q in the A-operand
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

// kv[b][o] = sum over the sample's tiles in a fixed order: 8 groups of threads take every 8th tile (four loads in
// flight each: the fold is a chain of L2 round trips, not bandwidth), then the groups are folded in order
__global__ __launch_bounds__(256) void k_ctx_kv_sum(TfArgs a) {
  const Seg& sg = a.seg[blockIdx.z];
  const int ol = threadIdx.x & 31, grp = threadIdx.x >> 5, o = blockIdx.x * 32 + ol, b = blockIdx.y;
  __shared__ float red[8][32];
  const float* p = sg.part + (size_t)b * sg.tiles * kKvFloats + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = grp;
  for (; t + 24 < sg.tiles; t += 32) {
    s0 += p[(size_t)t * kKvFloats];
    s1 += p[(size_t)(t + 8) * kKvFloats];
    s2 += p[(size_t)(t + 16) * kKvFloats];
    s3 += p[(size_t)(t + 24) * kKvFloats];
  }
  for (; t < sg.tiles; t += 8) s0 += p[(size_t)t * kKvFloats];
  red[grp][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0)
    sg.kv[(size_t)b * kKvFloats + o] = ((red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol])) +
                                       ((red[4][ol] + red[5][ol]) + (red[6][ol] + red[7][ol]));
}

// ------------------------------------------------------------------------------------------------ k_ctx_layer
// LDS (bytes): [0, 64K) planes of the MLP input: hi k-steps 0..15 = x, 16..31 = LN1(merge(attention)), then lo;
// [64K, 128K): first Q as float32 [chunk of 8 channels][token][8] (32K) and the planes of the attention output (2 x 16K),
// later the planes of the MLP hidden layer (2 x 32K); then the reduction scratch.
constexpr int kXmLo = 32 * 1024, kRegion2 = 64 * 1024, kAttHi = kRegion2 + 32 * 1024, kHidLo = 32 * 1024;
constexpr int kLayerWaves = 8;         // waves per k_ctx_layer workgroup (4 or 8; FM_CTX_LAYER_WAVES of a tuning build)
constexpr int layer_lds_bytes(int nw) { return 128 * 1024 + (2 * nw * 32 + 3 * nw * 32 + 2 * 32) * 4; }

// LayerNorm over the 256 channels of every token: this wave holds 32 NB of them (NB row blocks x 16 registers x 2 halves)
template <int NB, int NW>
__device__ __forceinline__ float fold_waves(const float* red, int r) {
  float t = (red[r] + red[32 + r]) + (red[64 + r] + red[96 + r]);
  if constexpr (NW == 8) t += (red[128 + r] + red[160 + r]) + (red[192 + r] + red[224 + r]);
  return t;
}
template <int NB, int NW>
__device__ __forceinline__ void layer_norm(f32x16 (&y)[NB], const float* __restrict__ gamma, const float* __restrict__ beta,
                                           float* red0, float* red1, int wv, int lane) {
  const int r = lane & 31, h = lane >> 5;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) s += y[i][g];
  s += other_half(s);
  if (h == 0) red0[wv * 32 + r] = s;
  __syncthreads();
  const float mean = fold_waves<NB, NW>(red0, r) * (1.0f / kD);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      y[i][g] -= mean;
      v = __builtin_fmaf(y[i][g], y[i][g], v);
    }
  v += other_half(v);
  if (h == 0) red1[wv * 32 + r] = v;
  __syncthreads();
  const float var = fold_waves<NB, NW>(red1, r) * (1.0f / kD);
  const float rstd = 1.0f / __builtin_sqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * (NB * wv + i) + 8 * q + 4 * h;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + ch), b4 = *reinterpret_cast<const f32x4*>(beta + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[i][4 * q + e] = __builtin_fmaf(y[i][4 * q + e] * rstd, g4[e], b4[e]);
    }
}

// NW waves per workgroup (4: one per SIMD, each owning a quarter of every layer's output channels; 8: two per SIMD, an
// eighth each - half the accumulators and half the weight ring per wave, so that one wave's matrix work runs while the
// other waits for its weight fragments)
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_ctx_layer(TfArgs a) {
  constexpr int NBQ = 8 / NW;        // 32-channel row blocks per wave of a 256-wide output (= heads per wave)
  constexpr int NBH = 16 / NW;       // ... of the 512-wide hidden layer
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const xm = lds;                                             // MLP input planes
  float* const qf = reinterpret_cast<float*>(lds + kRegion2);       // Q, float32
  char* const at = lds + kAttHi;                                    // attention output planes (hi, +16K lo)
  char* const hb = lds + kRegion2;                                  // hidden planes (hi, +32K lo)
  float* const red8 = reinterpret_cast<float*>(lds + 128 * 1024);
  float* const red0 = red8 + 2 * NW * 32;
  float* const red1 = red0 + NW * 32;
  float* const redm = red1 + NW * 32;
  float* const xinv = redm + NW * 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool s1 = (int)blockIdx.x >= a.tiles0;
  const Seg& sg = a.seg[s1 ? 1 : 0];
  const int wg = s1 ? (int)blockIdx.x - a.tiles0 : (int)blockIdx.x;
  const int b = wg / sg.tiles, tile = wg - b * sg.tiles, tok0 = tile * kTok;
#ifdef FM_DIAG_CTF
  const long long c0_ = __builtin_amdgcn_s_memtime();
  float* const dg = a.diag + ((size_t)blockIdx.x * NW + wv) * 16;
#endif
  const float* const hdr = reinterpret_cast<const float*>(a.w + kFragEnd);
  const float* const ln = hdr + kHdrLn;
  // x and (later) the LayerNorm-ed message share one scale per token, so that the MLP reads one operand: the scale
  // fits the larger of the token's |x| and the bound of the message (known at pack time)
  load_tile<2 * NW>(sg.x + (size_t)b * sg.L * kD, tok0, sg.L, xm, kXmLo, red8, xinv, xinv + 32, hdr[kHdrMsgBound], tid);
  CTF_STAMP(0);
  const float* const kvp = sg.kv + (size_t)b * kKvFloats;
  const float xi = xinv[r], xscale = xinv[32 + r];

  // ---- Q = elu(Wq x) + 1: this wave's two heads, float32 in LDS for the per-head products ----
  {
    f32x16 acc[NBQ] = {};
    gemm_stage<NBQ, 16, kXmLo>(a.w + kFragQ, NBQ * wv, xm, acc, lane);
    CTF_STAMP(1);
    const float f = hdr[kHdrWinv + 0] * xi;
    // padded query tokens (q_mask, attentions.py:35-36): Q = 0, hence a zero message (0 . KV / (0 + eps))
    const bool q_ok = !sg.mask || tok0 + r >= sg.L || sg.mask[(size_t)b * sg.L + tok0 + r] != 0;
#pragma unroll
    for (int i = 0; i < NBQ; ++i) {
      f32x4* p = reinterpret_cast<f32x4*>(qf) + (size_t)(4 * (NBQ * wv + i)) * 64 + r * 2 + h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = q_ok ? elu1(acc[i][4 * q + e] * f) : 0.f;
        p[q * 64] = v;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // ---- per head: (Q KV) / (Q . Ksum + eps) * S  (attentions.py:43-46), float32 MFMA; chunk j of 8 channels, step t:
  // half h contracts d = 8j + 4h + t ----
  f32x16 att[NBQ];
#pragma unroll
  for (int i = 0; i < NBQ; ++i) {
    const int hd = NBQ * wv + i;
    const f32x4* qa = reinterpret_cast<const f32x4*>(qf) + (size_t)(4 * hd) * 64 + r * 2 + h;
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
    const float zs = 1.0f / (den + 1e-6f) * sg.src_len;
#pragma unroll
    for (int g = 0; g < 16; ++g) att[i][g] = o[g] * zs;
  }
  CTF_STAMP(2);
  float as, ainv;
  token_scale<NBQ, NW>(att, redm, wv, lane, as, ainv);
#pragma unroll
  for (int i = 0; i < NBQ; ++i) {
#pragma unroll
    for (int g = 0; g < 16; ++g) att[i][g] *= as;
    store_planes(at, kPlane, NBQ * wv + i, att[i], lane);
  }
  __syncthreads();
  CTF_STAMP(3);
  // ---- merge + LayerNorm 1 -> k-steps 16..31 of the MLP input, in the token's x scale ----
  {
    f32x16 acc[NBQ] = {};
    gemm_stage<NBQ, 16, kPlane>(a.w + kFragM, NBQ * wv, at, acc, lane);
    CTF_STAMP(4);
    const float f = hdr[kHdrWinv + 3] * ainv;
#pragma unroll
    for (int i = 0; i < NBQ; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] *= f;
    layer_norm<NBQ, NW>(acc, ln, ln + kD, red0, red1, wv, lane);       // (its barriers also fence the reads of `at` above)
#pragma unroll
    for (int i = 0; i < NBQ; ++i) {
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] *= xscale;
      store_planes(xm, kXmLo, 8 + NBQ * wv + i, acc[i], lane);
    }
  }
  __syncthreads();
  CTF_STAMP(5);
  // ---- MLP: hidden = relu(W1 [x, msg]) ----
  float hinv;
  {
    f32x16 acc[NBH] = {};
    gemm_stage<NBH, 32, kXmLo>(a.w + kFragW1, NBH * wv, xm, acc, lane);
    CTF_STAMP(6);
    const float f = hdr[kHdrWinv + 4] * xi;
#pragma unroll
    for (int i = 0; i < NBH; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] = fmaxf(acc[i][g] * f, 0.f);
    float hs;
    token_scale<NBH, NW>(acc, redm, wv, lane, hs, hinv);
#pragma unroll
    for (int i = 0; i < NBH; ++i) {
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] *= hs;
      store_planes(hb, kHidLo, NBH * wv + i, acc[i], lane);
    }
  }
  __syncthreads();
  CTF_STAMP(7);
  // ---- W2 hidden -> LayerNorm 2 -> residual ----
  {
    f32x16 acc[NBQ] = {};
    gemm_stage<NBQ, 32, kHidLo>(a.w + kFragW2, NBQ * wv, hb, acc, lane);
    CTF_STAMP(8);
    const float f = hdr[kHdrWinv + 5] * hinv;
#pragma unroll
    for (int i = 0; i < NBQ; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][g] *= f;
    // the residual's x rows (L2) are requested before the LayerNorm, whose reductions hide the round trip
    const bool row_ok = tok0 + r < sg.L;
    const size_t row_off = ((size_t)b * sg.L + (row_ok ? tok0 + r : 0)) * kD;
    f32x4 xv[NBQ][4];
#pragma unroll
    for (int i = 0; i < NBQ; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[i][q] = *reinterpret_cast<const f32x4*>(sg.x + row_off + 32 * (NBQ * wv + i) + 8 * q + 4 * h);
    layer_norm<NBQ, NW>(acc, ln + 2 * kD, ln + 3 * kD, red0, red1, wv, lane);
    CTF_STAMP(9);
    if (row_ok) {
      float* const orow = sg.out + row_off;
#pragma unroll
      for (int i = 0; i < NBQ; ++i) {
        const int rbg = NBQ * wv + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = xv[i][q][e] + acc[i][4 * q + e];
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
  n += (t0 + t1) * 8 * 16;
#endif
  return n;
}

}  // namespace

extern "C" size_t fm_coarse_tf_packed_bytes(int n_layers) {
  return n_layers > 0 && n_layers <= kMaxLayers ? (size_t)n_layers * kLayerBytes : 0;
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
  if ((uintptr_t)packed & 15) return FM_E_WORKSPACE;
  for (int l = 0; l < n_layers; ++l) {
    if (!layers[l]) return FM_E_NULL;
    for (int i = 0; i < 10; ++i)
      if (!layers[l][i]) return FM_E_NULL;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t off[6] = {kFragQ, kFragK, kFragV, kFragM, kFragW1, kFragW2};
  const int n_out[6] = {256, 256, 256, 256, 512, 256}, n_in[6] = {256, 256, 256, 256, 512, 512};
  for (int l = 0; l < n_layers; ++l) {
    char* dst = (char*)packed + (size_t)l * kLayerBytes;
    float* hdr = (float*)(dst + kFragEnd);
    hipError_t e = hipMemsetAsync(hdr, 0, kHdrFloats * 4, st);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < 6; ++i) {
      const int n = n_out[i] * n_in[i];
      unsigned* amax = (unsigned*)(hdr + kHdrAbsmax + i);
      hipLaunchKernelGGL(k_ctx_absmax, dim3(64), dim3(256), 0, st, layers[l][i], n, amax);
      hipLaunchKernelGGL(k_ctx_pack16, dim3((n / 8 + 255) / 256), dim3(256), 0, st, layers[l][i], n_out[i], n_in[i], amax,
                         (half8*)(dst + off[i]), hdr + kHdrWinv + i);
    }
    hipLaunchKernelGGL(k_ctx_pack_ln, dim3(1), dim3(256), 0, st, layers[l][6], layers[l][7], layers[l][8], layers[l][9], hdr);
  }
  return (int)hipGetLastError();
}

// layer_kinds[l]: 0 = 'self', 1 = 'cross' (transformer.py:88-95).  out0 / out1 [N,L,256] / [N,S,256] must not overlap
// the inputs or each other (the first layer reads feat*, every later one updates out* in place): FM_E_UNSUPPORTED.
extern "C" int fm_coarse_transformer(const float* feat0, const float* feat1, int N, int L, int S, int C, int nhead,
                                     const int* layer_kinds, int n_layers, const void* packed, void* workspace,
                                     size_t workspace_bytes, float* out0, float* out1, void* stream) {
  return fm_coarse_transformer_masked(feat0, feat1, nullptr, nullptr, N, L, S, C, nhead, layer_kinds, n_layers, packed, workspace,
                                      workspace_bytes, out0, out1, stream);
}

// The same with the reference's padding masks (transformer.py:78-96 mask0 / mask1, attentions.py:35-40): mask0 [N, L],
// mask1 [N, S] bytes (1 = a real token, 0 = padding; torch.bool storage), either may be NULL.
extern "C" int fm_coarse_transformer_masked(const float* feat0, const float* feat1, const unsigned char* mask0,
                                            const unsigned char* mask1, int N, int L, int S, int C, int nhead,
                                            const int* layer_kinds, int n_layers, const void* packed, void* workspace,
                                            size_t workspace_bytes, float* out0, float* out1, void* stream) {
  if (!feat0 || !feat1 || !layer_kinds || !packed || !workspace || !out0 || !out1) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  if (C != kD || nhead != kH || n_layers <= 0 || n_layers > kMaxLayers) return FM_E_UNSUPPORTED;
  for (int l = 0; l < n_layers; ++l)
    if (layer_kinds[l] != 0 && layer_kinds[l] != 1) return FM_E_UNSUPPORTED;
  {   // no output may overlap an input or the other output: a 'self' layer updates both images in one launch and the
      // first layer reads feat* while it writes out* (byte ranges, not just equal pointers)
    const uintptr_t n0 = (uintptr_t)N * L * kD * 4, n1 = (uintptr_t)N * S * kD * 4;
    auto overlap = [](const void* a, uintptr_t na, const void* b, uintptr_t nb) {
      return (uintptr_t)a < (uintptr_t)b + nb && (uintptr_t)b < (uintptr_t)a + na;
    };
    if (overlap(out0, n0, out1, n1) || overlap(out0, n0, feat0, n0) || overlap(out0, n0, feat1, n1) ||
        overlap(out1, n1, feat0, n0) || overlap(out1, n1, feat1, n1))
      return FM_E_UNSUPPORTED;
  }
  if (workspace_bytes < ws_floats(N, L, S) * 4 || ((uintptr_t)workspace & 15) || ((uintptr_t)packed & 15))
    return FM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long set_kv = 0, set_layer = 0;
  hipError_t e = ensure_dynamic_lds(&k_ctx_kv, kKvLdsBytes, &set_kv);
  if (e != hipSuccess) return (int)e;
  int layer_waves = kLayerWaves;
#ifdef FM_TUNE_ENV
  if (const char* ev = getenv("FM_CTX_LAYER_WAVES")) layer_waves = atoi(ev) == 4 ? 4 : 8;
#endif
  static unsigned long long set_layer4 = 0;
  e = layer_waves == 8 ? ensure_dynamic_lds(&k_ctx_layer<8>, layer_lds_bytes(8), &set_layer)
                       : ensure_dynamic_lds(&k_ctx_layer<4>, layer_lds_bytes(4), &set_layer4);
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
  const unsigned char* masks[2] = {mask0, mask1};

  // one encoder layer on `nseg` (image, source) pairs: x[img[s]] <- layer(x[img[s]], x[src[s]])
  auto run = [&](const char* w, int nseg, const int* img, const int* src) -> hipError_t {
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
      g.mask = masks[src[s]];
      tiles_kv += N * g.tiles;
    }
    a.tiles0 = N * a.seg[0].tiles;
    hipLaunchKernelGGL(k_ctx_kv, dim3(tiles_kv), dim3(256), kKvLdsBytes, st, a);
    hipLaunchKernelGGL(k_ctx_kv_sum, dim3(kKvFloats / 32, N, nseg), dim3(256), 0, st, a);
    // query side: the tokens that are updated
    int tiles_x = 0;
    for (int s = 0; s < nseg; ++s) {
      Seg& g = a.seg[s];
      g.x = cur[img[s]]; g.out = out[img[s]]; g.part = nullptr; g.kv = kv[src[s]];
      g.L = len[img[s]]; g.tiles = tl[img[s]]; g.src_len = (float)len[src[s]];
      g.mask = masks[img[s]];
      tiles_x += N * g.tiles;
    }
    a.tiles0 = N * a.seg[0].tiles;
    if (layer_waves == 8) hipLaunchKernelGGL(k_ctx_layer<8>, dim3(tiles_x), dim3(512), layer_lds_bytes(8), st, a);
    else hipLaunchKernelGGL(k_ctx_layer<4>, dim3(tiles_x), dim3(256), layer_lds_bytes(4), st, a);
    for (int s = 0; s < nseg; ++s) cur[img[s]] = out[img[s]];
    return hipGetLastError();
  };

  for (int l = 0; l < n_layers; ++l) {
    const char* w = (const char*)packed + (size_t)l * kLayerBytes;
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
