// Fine-level context layers: the reference's LocalFeatureTransformer on the matched windows (SURVEY.md 8(f) row 1;
// network/net.py:79-80, network/module/transformer.py:34-57,78-96, network/module/attentions.py:19-46) with the
// default fine configuration: d_model 64, 8 heads, layer_names ['self', 'cross'], linear attention, no masks.
//
//   layer(x, src):  q = x Wq^T, k = src Wk^T, v = src Wv^T                       (bias-free Linear layers)
//                   Q = elu(q)+1, K = elu(k)+1, per head: KV = K^T (v/S), Z = 1/(Q.sum_s K + 1e-6)
//                   msg = (Q KV) Z S -> merge -> LayerNorm1 -> MLP([x | msg]): 128 -> 128 -> ReLU -> 64 -> LayerNorm2
//                   return x + msg
//   self : f0 = layer0(f0, f0); f1 = layer0(f1, f1)      cross: f0 = layer1(f0, f1); f1 = layer1(f1, f0_new)
//
// ONE WAVE PER MATCH (four matches per workgroup, which share the weight fragments staged through LDS by LDS-DMA:
// every wave pulling the 640 KB of fragments of the four layer calls by itself made the kernel latency-bound on
// those loads, 800 us at 640x480), and every activation stays in registers in MFMA accumulator layout from the window load to
// the window store.  The trick is to compute each product in the orientation whose OUTPUT feeds the next product
// as an operand without lane movement (an accumulator tile has its column on the lane and its rows in the 16
// registers, so it is directly the operand of a product that sums over its ROW index):
//   T layout: features in registers, tokens on lanes   (result of  W . X^T ; operand B of the next W . X^T)
//   N layout: tokens in registers, features on lanes   (result of  S . W^T ; K and V, which are summed over tokens)
//     q^T   = Wq . x^T                 (T)      k, v = src . W^T            (N: the same src registers as operand A)
//     KV    = K^T . V                  (sum over tokens = row index of both N tiles; only the diagonal 32x32 tiles
//                                       hold head blocks, masked to the 8x8 blocks of the 8 heads)
//     msg^T = KV^T . Q^T               (sum over d = row index of KV and of Q^T)          -> T
//     den^T = Kd . Q^T                 (Kd[head][d] = sum_s K[s][d] inside the head, else 0) -> Z per token and head
//     merge, MLP                       (W . X^T, T -> T);  LayerNorm over features = over registers + one half swap
// Products are float32-equivalent: both operands are split x = hi + lo (float16 each, 22 mantissa bits) and
// hi*hi + lo*hi + hi*lo is accumulated in float32 on the matrix cores (v_mfma_f32_32x32x16_f16).  The weights are
// pre-split and pre-permuted into operand fragments by fm_fine_tf_pack_weights: an accumulator-derived fragment of
// k-step s holds k = 16 s + 8 (j >> 2) + 4 h + (j & 3) in element j of lane half h, so the weight fragments use the
// same order.
#include "fm_internal.h"

namespace fm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// packed weights of one layer, in half8 fragments [plane hi/lo][out tile][k-step][lane]
constexpr int kTfFragQ = 0;                    // q_proj  [64 x 64]: 2 out tiles x 4 k-steps
constexpr int kTfFragK = kTfFragQ + 2 * 4;
constexpr int kTfFragV = kTfFragK + 2 * 4;
constexpr int kTfFragM = kTfFragV + 2 * 4;     // merge
constexpr int kTfFrag1 = kTfFragM + 2 * 4;     // mlp.0   [128 x 128]: 4 out tiles x 8 k-steps
constexpr int kTfFrag2 = kTfFrag1 + 4 * 8;     // mlp.2   [64 x 128]: 2 out tiles x 8 k-steps
constexpr int kTfFrags = kTfFrag2 + 2 * 8;     // 80 fragments of 64 lanes x 8 halves per plane
constexpr int kTfLayerHalf8 = 2 * kTfFrags * 64;                  // hi plane then lo plane
constexpr int kTfLayerFloats = 4 * 64;                            // norm1.weight, norm1.bias, norm2.weight, norm2.bias

struct Tile { f32x16 t[2][2]; };        // [row tile][column tile] of a 64 x 64 matrix in accumulator layout

// Operands carry exact power-of-two scales: the lo half of a value below 2^-3 would otherwise be a float16
// SUBNORMAL (|lo| ~ 2^-12 |x| < 2^-14), which the matrix cores flush - the weights of a 64..128-wide Linear layer
// are all below that (measured: 3e-4 instead of 1e-6).  Every tile that feeds a product is therefore KEPT in the
// operand scale (kActScale times its value) from the window load to the window store: the accumulators are scaled
// back by the weight scale only, LayerNorm runs on scaled values with a scaled epsilon (it is scale invariant
// otherwise), relu commutes with the scale, and elu(x)+1 folds it into its constants.
constexpr float kActScale = 256.f;          // activations: |x| < 256 stays inside float16
constexpr float kWgtScale = 4096.f;         // weights (xavier bound <= 0.31)
constexpr float kSumScale = 32.f;           // sum_s K (up to ~1e3)

typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = hi + lo in float16 (round-toward-zero packs: the remainder x - hi is exact in float32, and the scheme only
// needs hi + lo = x to 22 bits, not nearest rounding): 3 VALU operations per element.  The tile is ALREADY in the
// operand scale (see above), so there is no multiply here.
__device__ __forceinline__ void split8(const f32x16& a, int half, half8& hi, half8& lo) {
  u32x4 uh, ul;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = a[8 * half + 2 * p], x1 = a[8 * half + 2 * p + 1];
    const fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const fp16x2 l2 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h2[0], x1 - (float)h2[1]);
    uh[p] = __builtin_bit_cast(unsigned, h2);
    ul[p] = __builtin_bit_cast(unsigned, l2);
  }
  hi = __builtin_bit_cast(half8, uh);
  lo = __builtin_bit_cast(half8, ul);
}
// acc += A.B with both operands split (float32-equivalent product)
__device__ __forceinline__ void mma3(f32x16& acc, const half8& ah, const half8& al, const half8& bh, const half8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
}
__device__ __forceinline__ void rescale(f32x16& a, float f) {
#pragma unroll
  for (int g = 0; g < 16; ++g) a[g] *= f;
}
__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
  for (int g = 0; g < 16; ++g) a[g] = 0.f;
}
__device__ __forceinline__ float swap_halves_add(float v) {       // v(lane) + v(lane ^ 32)
  float p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return p + q;
}
__device__ __forceinline__ float other_half(float v) {            // v(lane ^ 32)
  float p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return (threadIdx.x & 32) ? p : q;
}

// Y^T[OT*32 x 64 tokens] (+)= W[rows ot0*32.., columns s0*16..] . X^T : T layout in, T layout out.  rows[i] = the
// i-th 32-feature row tile of the source (its two token tiles), KS = 2 * number of row tiles; wf = the matrix'
// fragments in LDS (hi plane wf, lo plane wfl), KSW = k-steps of the whole matrix.  The accumulators keep the
// operand scale until `finish` (so that a product can be accumulated in pieces).
template <int OT, int KS, int KSW>
__device__ __forceinline__ void gemm_T(f32x16 (&out)[OT][2], const f32x16 (*const (&rows)[KS / 2])[2],
                                       const half8* wf, const half8* wfl, int ot0, int s0, bool first, bool finish, int lane) {
  if (first) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { zero(out[ot][0]); zero(out[ot][1]); }
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    half8 bh[2], bl[2];
    split8((*rows[s >> 1])[0], s & 1, bh[0], bl[0]);
    split8((*rows[s >> 1])[1], s & 1, bh[1], bl[1]);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const int fi = ((ot0 + ot) * KSW + s0 + s) * 64 + lane;
      const half8 wh = wf[fi], wl = wfl[fi];
      mma3(out[ot][0], wh, wl, bh[0], bl[0]);
      mma3(out[ot][1], wh, wl, bh[1], bl[1]);
    }
  }
  if (finish) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { rescale(out[ot][0], 1.0f / kWgtScale); rescale(out[ot][1], 1.0f / kWgtScale); }
  }
}

// K[64 tokens x 64] = S . Wk^T and V = S . Wv^T : the T-layout registers of S as operand A (one split serves both
// products), N layout out (tokens in registers, features on lanes)
__device__ __forceinline__ void gemm_N2(Tile& outk, Tile& outv, const Tile& src, const half8* wk, const half8* wkl,
                                        const half8* wv, const half8* wvl, int lane) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) { zero(outk.t[rt][0]); zero(outk.t[rt][1]); zero(outv.t[rt][0]); zero(outv.t[rt][1]); }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    half8 ah[2], al[2];
    split8(src.t[s >> 1][0], s & 1, ah[0], al[0]);       // token tile 0
    split8(src.t[s >> 1][1], s & 1, ah[1], al[1]);       // token tile 1
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      const int fi = (ot * 4 + s) * 64 + lane;
      const half8 kh = wk[fi], kl = wkl[fi];
      mma3(outk.t[0][ot], ah[0], al[0], kh, kl);
      mma3(outk.t[1][ot], ah[1], al[1], kh, kl);
      const half8 vh = wv[fi], vl = wvl[fi];
      mma3(outv.t[0][ot], ah[0], al[0], vh, vl);
      mma3(outv.t[1][ot], ah[1], al[1], vh, vl);
    }
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { rescale(outk.t[rt][ct], 1.0f / kWgtScale); rescale(outv.t[rt][ct], 1.0f / kWgtScale); }
}

// elu(x) + 1 of a tile in the operand scale, result in the operand scale: xs = A x -> A (x > 0 ? x + 1 : exp(x))
__device__ __forceinline__ float elu1_scaled(float xs) {
  constexpr float kL2A = 8.0f;                          // log2(kActScale)
  return xs > 0.f ? xs + kActScale : __builtin_amdgcn_exp2f(__builtin_fmaf(xs, kLog2e / kActScale, kL2A));
}

// LayerNorm over the 64 features of every token (T layout: this lane's 32 features + the other half's 32), eps 1e-5;
// input and output in the operand scale (gamma and beta are packed pre-multiplied by kActScale)
__device__ __forceinline__ void layer_norm_T(f32x16 (&y)[2][2], const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int h) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float s = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 16; ++g) s += y[rt][ct][g];
    const float mean = swap_halves_add(s) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 16; ++g) { const float d = y[rt][ct][g] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(swap_halves_add(v) * (1.0f / 64.0f) + 1e-5f * kActScale * kActScale);   // scaled eps
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 32 * rt + 8 * q + 4 * h);
        const float4 be = *reinterpret_cast<const float4*>(beta + 32 * rt + 8 * q + 4 * h);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) y[rt][ct][4 * q + e] = (y[rt][ct][4 * q + e] - mean) * rstd * gg[e] + bb[e];
      }
  }
}

// x <- x + LN2(MLP([x | LN1(merge(attention(x, src)))]))      (transformer.py:34-57)
// 1 KiB fragment blocks global -> LDS by LDS-DMA (no registers): fragment f of the block goes to dst + f KiB; the four
// waves take every fourth fragment.  Completion: stage_wait().
__device__ __forceinline__ void stage_frags(char* dst, const half8* src, int nfrags, int wv, int lane) {
  for (int f = wv; f < nfrags; f += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
}
__device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// LDS map of the staged weights: region A (64 KiB) = the attention matrices q, k, v, merge [hi 32 frags | lo 32 frags],
// later the second MLP matrix [hi 16 | lo 16]; region B (64 KiB) = the first MLP matrix [hi 32 | lo 32].
constexpr int kRegionBytes = 64 * 1024;

template <int WW>
__device__ __forceinline__ void encoder_layer(Tile& x, const Tile& src, const half8* __restrict__ wl, bool load_w1,
                                              const float* __restrict__ ln, char* lds, float* ksum_lds, int wv, int lane) {
  const int r = lane & 31, h = lane >> 5;
  // every wave of the workgroup is past the previous layer's use of both regions (barrier), then the weights of this
  // layer arrive: 32 + 32 fragments of the attention matrices, and - unless the previous call left it there - the
  // first MLP matrix
  __syncthreads();
  stage_frags(lds, wl + kTfFragQ * 64, 32, wv, lane);
  stage_frags(lds + 32 * 1024, wl + kTfFrags * 64 + kTfFragQ * 64, 32, wv, lane);
  if (load_w1) {
    stage_frags(lds + kRegionBytes, wl + kTfFrag1 * 64, 32, wv, lane);
    stage_frags(lds + kRegionBytes + 32 * 1024, wl + kTfFrags * 64 + kTfFrag1 * 64, 32, wv, lane);
  }
  stage_wait();
  const half8* const la = reinterpret_cast<const half8*>(lds);                       // region A, hi plane
  const half8* const lal = reinterpret_cast<const half8*>(lds + 32 * 1024);          // region A, lo plane
  const half8* const lb = reinterpret_cast<const half8*>(lds + kRegionBytes);
  const half8* const lbl = reinterpret_cast<const half8*>(lds + kRegionBytes + 32 * 1024);
  // ---- projections ----
  Tile q;                                   // q^T, T layout
  {
    const f32x16 (*const rows[2])[2] = {&x.t[0], &x.t[1]};
    gemm_T<2, 4, 4>(q.t, rows, la + kTfFragQ * 64, lal + kTfFragQ * 64, 0, 0, true, true, lane);
  }
  Tile k, v;                                // N layout: token = 32 rt + (g&3) + 8 (g>>2) + 4 h, feature on the lane
  gemm_N2(k, v, src, la + kTfFragK * 64, lal + kTfFragK * 64, la + kTfFragV * 64, lal + kTfFragV * 64, lane);
  // feature maps; padded tokens (>= WW) must not enter the sums over tokens
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        q.t[rt][ct][g] = elu1_scaled(q.t[rt][ct][g]);
        const bool tok_ok = 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * h < WW;
        k.t[rt][ct][g] = tok_ok ? elu1_scaled(k.t[rt][ct][g]) : 0.f;
        v.t[rt][ct][g] = v.t[rt][ct][g] * (1.0f / (float)WW);      // values / S (attentions.py:41-42)
      }
  // ---- sum_s K[s][d] per feature d (on the lane): over this lane's token registers, then the other half ----
  {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float s = 0.f;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 16; ++g) s += k.t[rt][ct][g];
      s = swap_halves_add(s);
      if (h == 0) ksum_lds[32 * ct + r] = s;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- KV = K^T . V : only the two diagonal 32 x 32 tiles hold head blocks ----
  f32x16 kv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    zero(kv[dt]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {           // k-steps over the 64 (padded) tokens
      half8 ah, al, bh, bl;
      split8(k.t[s >> 1][dt], s & 1, ah, al);
      split8(v.t[s >> 1][dt], s & 1, bh, bl);
      mma3(kv[dt], ah, al, bh, bl);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g)            // keep d / 8 == v / 8 (rows d = (g&3) + 8 (g>>2) + 4 h, column v = r)
      kv[dt][g] = ((g >> 2) != (r >> 3)) ? 0.f : kv[dt][g] * (1.0f / kActScale);     // back to the operand scale
  }
  // ---- msg^T = KV^T . Q^T and den^T = Kd . Q^T (both sum over d, the row index of KV and of Q^T) ----
  Tile msg;
  f32x16 den[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) { zero(msg.t[0][ct]); zero(msg.t[1][ct]); zero(den[ct]); }
#pragma unroll
  for (int s = 0; s < 4; ++s) {             // k-steps over d
    const int dt = s >> 1;
    half8 ah, al;                           // KV rows d as operand A (X^T . B form)
    split8(kv[dt], s & 1, ah, al);
    // Kd fragment: row = head r (< 8), element j = sum_s K of feature d = 16 s + 8 (j>>2) + 4 h + (j&3) if in head r
    half8 dh, dl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      const float val = (r == (d >> 3)) ? ksum_lds[d] * (kSumScale / kActScale) : 0.f;     // ksum_lds is in the operand scale
      const _Float16 hh = (_Float16)val;
      dh[j] = hh;
      dl[j] = (_Float16)(val - (float)hh);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      half8 bh, bl;
      split8(q.t[dt][ct], s & 1, bh, bl);
      mma3(msg.t[dt][ct], ah, al, bh, bl);  // KV[dt] only reaches output rows v in tile dt
      mma3(den[ct], dh, dl, bh, bl);
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    rescale(msg.t[0][ct], 1.0f / kActScale);
    rescale(msg.t[1][ct], 1.0f / kActScale);
  }
  // Z[token][head] = 1 / (den + eps): den rows 0..3 sit in registers 0..3 of half 0, rows 4..7 in half 1
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float z[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float mine = den[ct][e] * (1.0f / (kSumScale * kActScale)), theirs = other_half(den[ct][e]) * (1.0f / (kSumScale * kActScale));   // true scale
      z[e] = h ? theirs : mine;             // heads 0..3
      z[4 + e] = h ? mine : theirs;         // heads 4..7
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (float)WW / (z[e] + 1e-6f);     // ... * S (attentions.py:46)
    // msg^T rows v = 32 rt + (g&3) + 8 (g>>2) + 4 h: head = 4 rt + (g >> 2)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 16; ++g) msg.t[rt][ct][g] *= z[4 * rt + (g >> 2)];
  }
  // ---- merge + LayerNorm1 ----
  Tile m1;
  {
    const f32x16 (*const rows[2])[2] = {&msg.t[0], &msg.t[1]};
    gemm_T<2, 4, 4>(m1.t, rows, la + kTfFragM * 64, lal + kTfFragM * 64, 0, 0, true, true, lane);
  }
  // region A is free once every wave has finished its merge product: the second MLP matrix takes its place while
  // the first MLP product runs out of region B
  __syncthreads();
  stage_frags(lds, wl + kTfFrag2 * 64, 16, wv, lane);
  stage_frags(lds + 16 * 1024, wl + kTfFrags * 64 + kTfFrag2 * 64, 16, wv, lane);
  layer_norm_T(m1.t, ln, ln + 64, h);
  // ---- MLP on [x | msg] + LayerNorm2 + residual (one split of x and msg serves all four hidden row tiles) ----
  f32x16 hid[4][2];
  {
    const f32x16 (*const rows[4])[2] = {&x.t[0], &x.t[1], &m1.t[0], &m1.t[1]};
    gemm_T<4, 8, 8>(hid, rows, lb, lbl, 0, 0, true, true, lane);
  }
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) hid[ot][ct][g] = fmaxf(hid[ot][ct][g], 0.f);
  stage_wait();
  Tile m2;
  {
    const f32x16 (*const rows[4])[2] = {&hid[0], &hid[1], &hid[2], &hid[3]};
    gemm_T<2, 8, 8>(m2.t, rows, la, reinterpret_cast<const half8*>(lds + 16 * 1024), 0, 0, true, true, lane);
  }
  layer_norm_T(m2.t, ln + 128, ln + 192, h);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) x.t[rt][ct][g] += m2.t[rt][ct][g];
}

// window [WW, 64] (token-major) <-> T layout: lane (token r of tile ct, half h) holds features 32 rt + 8 q + 4 h + 0..3
template <int WW>
__device__ __forceinline__ void load_window_T(Tile& x, const float* __restrict__ win, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int tok = 32 * ct + r;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok < WW) v = *reinterpret_cast<const float4*>(win + tok * 64 + 32 * rt + 8 * q + 4 * h);
        x.t[rt][ct][4 * q] = v.x * kActScale; x.t[rt][ct][4 * q + 1] = v.y * kActScale;      // into the operand scale
        x.t[rt][ct][4 * q + 2] = v.z * kActScale; x.t[rt][ct][4 * q + 3] = v.w * kActScale;
      }
  }
}
template <int WW>
__device__ __forceinline__ void store_window_T(const Tile& x, float* __restrict__ win, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int tok = 32 * ct + r;
    if (tok >= WW) continue;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(win + tok * 64 + 32 * rt + 8 * q + 4 * h) =
            make_float4(x.t[rt][ct][4 * q] * (1.0f / kActScale), x.t[rt][ct][4 * q + 1] * (1.0f / kActScale),
                        x.t[rt][ct][4 * q + 2] * (1.0f / kActScale), x.t[rt][ct][4 * q + 3] * (1.0f / kActScale));
  }
}

template <int WW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_fine_tf(const float* __restrict__ win0, const float* __restrict__ win1, int m_max,
               const int32_t* __restrict__ d_count, const half8* __restrict__ wpack, const float* __restrict__ lnp,
               float* __restrict__ out0, float* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // 128 KiB of staged weight fragments (two regions)
  __shared__ float ksum[4][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if ((int)blockIdx.x * 4 >= M) return;                           // uniform: nothing left for this workgroup
  // one wave per match, four matches per workgroup (they share the staged weights); a wave beyond the last match
  // works on the last one again and does not store
  const int m = min((int)blockIdx.x * 4 + wv, M - 1);
  const bool store = (int)blockIdx.x * 4 + wv < M;
  Tile f0, f1;
  load_window_T<WW>(f0, win0 + (long)m * WW * 64, lane);
  load_window_T<WW>(f1, win1 + (long)m * WW * 64, lane);
  // 'self' (transformer.py:89-91): the same layer on each image by itself
  encoder_layer<WW>(f0, f0, wpack, true, lnp, lds, ksum[wv], wv, lane);
  encoder_layer<WW>(f1, f1, wpack, false, lnp, lds, ksum[wv], wv, lane);
  // 'cross' (:92-94): feat0 from feat1, then feat1 from the UPDATED feat0
  encoder_layer<WW>(f0, f1, wpack + kTfLayerHalf8, true, lnp + kTfLayerFloats, lds, ksum[wv], wv, lane);
  encoder_layer<WW>(f1, f0, wpack + kTfLayerHalf8, false, lnp + kTfLayerFloats, lds, ksum[wv], wv, lane);
  if (store) {
    store_window_T<WW>(f0, out0 + (long)m * WW * 64, lane);
    store_window_T<WW>(f1, out1 + (long)m * WW * 64, lane);
  }
}

// One weight matrix W [OUT x IN] (row-major, nn.Linear.weight) -> operand fragments [out tile][k-step][lane] x 8 halves,
// hi plane at dst, lo plane at dst + kTfFrags*64: element j of lane (r, h) = W[32 ot + r][16 s + 8 (j>>2) + 4 h + (j&3)]
__global__ __launch_bounds__(256) void k_tf_pack(const float* __restrict__ w, int out_f, int in_f, half8* __restrict__ dst) {
  const int ks = in_f / 16;
  const int idx = blockIdx.x * 256 + threadIdx.x;      // (ot, s, lane)
  if (idx >= out_f / 32 * ks * 64) return;
  const int lane = idx & 63, s = (idx >> 6) % ks, ot = (idx >> 6) / ks;
  const int r = lane & 31, h = lane >> 5;
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = w[(long)(32 * ot + r) * in_f + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)] * kWgtScale;
    const _Float16 hh = (_Float16)x;
    hi[j] = hh;
    lo[j] = (_Float16)(x - (float)hh);
  }
  dst[idx] = hi;
  dst[kTfFrags * 64 + idx] = lo;
}

__global__ void k_tf_scale_copy(const float* __restrict__ src, float* __restrict__ dst, float scale) {
  dst[threadIdx.x] = src[threadIdx.x] * scale;
}

}  // namespace fm

using namespace fm;

extern "C" size_t fm_fine_tf_packed_bytes(void) { return 2 * ((size_t)kTfLayerHalf8 * 16 + kTfLayerFloats * 4); }

// layer_weights[l] for l = 0 ('self'), 1 ('cross'): pointers to q_proj, k_proj, v_proj, merge [64,64], mlp.0 [128,128],
// mlp.2 [64,128], norm1.weight, norm1.bias, norm2.weight, norm2.bias [64]  (10 device pointers per layer)
extern "C" int fm_fine_tf_pack_weights(const float* const* layer0, const float* const* layer1, void* packed, void* stream) {
  if (!layer0 || !layer1 || !packed) return FM_E_NULL;
  for (int i = 0; i < 10; ++i)
    if (!layer0[i] || !layer1[i]) return FM_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  half8* frag = (half8*)packed;
  float* ln = (float*)((char*)packed + 2 * (size_t)kTfLayerHalf8 * 16);
  const int base[6] = {kTfFragQ, kTfFragK, kTfFragV, kTfFragM, kTfFrag1, kTfFrag2};
  const int outf[6] = {64, 64, 64, 64, 128, 64}, inf[6] = {64, 64, 64, 64, 128, 128};
  for (int l = 0; l < 2; ++l) {
    const float* const* w = l ? layer1 : layer0;
    for (int i = 0; i < 6; ++i) {
      const int n = outf[i] / 32 * (inf[i] / 16) * 64;
      hipLaunchKernelGGL(k_tf_pack, dim3((n + 255) / 256), dim3(256), 0, st, w[i], outf[i], inf[i],
                         frag + (size_t)l * kTfLayerHalf8 + base[i] * 64);
    }
    for (int i = 0; i < 4; ++i)        // gamma and beta pre-multiplied by the operand scale
      hipLaunchKernelGGL(k_tf_scale_copy, dim3(1), dim3(64), 0, st, w[6 + i], ln + l * kTfLayerFloats + 64 * i, kActScale);
  }
  return (int)hipGetLastError();
}

extern "C" int fm_fine_transformer(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                                   const void* packed, float* out0, float* out1, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!win0 || !win1 || !packed || !out0 || !out1) return FM_E_NULL;
  if (m_max < 0) return FM_E_SHAPE;
  if (Cf != 64 || (WW != 25 && WW != 49)) return FM_E_UNSUPPORTED;
  const half8* frag = (const half8*)packed;
  const float* ln = (const float*)((const char*)packed + 2 * (size_t)kTfLayerHalf8 * 16);
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (m_max + 3) / 4;
  const int smem = 2 * kRegionBytes;
  static unsigned long long set49 = 0, set25 = 0;
  if (WW == 49) {
    hipError_t e = ensure_dynamic_lds(&k_fine_tf<49>, smem, &set49);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_fine_tf<49>, dim3(blocks), dim3(256), smem, st, win0, win1, m_max, d_count, frag, ln, out0, out1);
  } else {
    hipError_t e = ensure_dynamic_lds(&k_fine_tf<25>, smem, &set25);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_fine_tf<25>, dim3(blocks), dim3(256), smem, st, win0, win1, m_max, d_count, frag, ln, out0, out1);
  }
  return (int)hipGetLastError();
}
