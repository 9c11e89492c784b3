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
// ONE WAVE PER MATCH, eight matches per workgroup, which share the weight fragments staged through LDS by LDS-DMA
// (every wave pulling the 640 KB of fragments of the four layer calls by itself made the kernel latency-bound on
// those loads).  A wave works on one 32-token SLICE of a window at a time and keeps it in registers in MFMA
// accumulator layout from its load to its store; a layer call x <- layer(x, src) is a kv phase over the slices of src
// (K, V projections, KV += K^T V, sum K) and an update phase over the slices of x, the windows re-read in between (L2).
// The first version kept both windows of a match in registers (128 of 512) and spilled: its later layer calls took
// 85-120 k cycles against 57 k for the first; with slices the live state fits 256 registers (two waves per SIMD, no
// scratch) and a 5 x 5 window is one slice instead of a padded pair: 693 -> 393 us at 3800 matches (W = 7).
// The trick that keeps activations in registers is to compute each product in the orientation whose OUTPUT feeds the
// next product as an operand without lane movement (an accumulator tile has its column on the lane and its rows in the
// 16 registers, so it is directly the operand of a product that sums over its ROW index):
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


// Operands carry exact power-of-two scales: the lo half of a value below 2^-3 would otherwise be a float16
// SUBNORMAL (|lo| ~ 2^-12 |x| < 2^-14), which the matrix cores flush - the weights of a 64..128-wide Linear layer
// are all below that (measured: 3e-4 instead of 1e-6).  Every tile that feeds a product is therefore KEPT in the
// operand scale (the wave's ActScale times its value) from the window load to the window store: the accumulators are scaled
// back by the weight scale only, LayerNorm runs on scaled values with a scaled epsilon (it is scale invariant
// otherwise), relu commutes with the scale, and elu(x)+1 folds it into its constants.
constexpr float kLog2ActScale = 8.f;        // activations: the first attempt's scale 2^8 (|x| < 256 stays inside float16; see ActScale)
constexpr float kWgtScale = 4096.f;         // weights (xavier bound <= 0.31)
constexpr float kSumScale = 32.f;           // sum_s K (up to ~1e3)

// The activation scale of a wave (= of its match): 2^8 at first; a match whose operands leave float16 at that scale
// is recomputed with 2^4, 2^0, 2^-4 (k_fine_tf: the four layer calls start again from the input windows).  At a
// smaller scale the lo halves of values below 2^-3 / scale are flushed (float16 subnormals), i.e. elements three
// orders of magnitude below the ones that forced the scale down lose their last 11 bits - against an answer that was
// not available at all before (the call reported FM_DEV_RANGE and the module ran its float32 layers, 20x slower).
struct ActScale {
  float a, inv, log2a;       // scale, 1 / scale, log2(scale)
};
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = hi + lo in float16 (round-toward-zero packs: the remainder x - hi is exact in float32, and the scheme only
// needs hi + lo = x to 22 bits, not nearest rounding): 3 VALU operations per element.  The tile is ALREADY in the
// operand scale (see above), so there is no multiply here.
// RANGE: a value beyond float16 at the wave's scale (|x| * scale > 65504: a window value, a projection or a
// hidden-layer value above 255.9) would be clamped silently by the round-toward-zero pack.  `amax` follows the largest
// magnitude that ever went into an operand (one v_max3 per pair of elements); the kernel reports FM_DEV_RANGE through
// its status word when it exceeded the float16 range, and the caller falls back to float32 layers.
__device__ __forceinline__ void split8(const f32x16& a, int half, half8& hi, half8& lo, float& amax) {
  u32x4 uh, ul;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = a[8 * half + 2 * p], x1 = a[8 * half + 2 * p + 1];
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x0), "v"(x1));
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

// elu(x) + 1 of a tile in the operand scale, result in the operand scale: xs = A x -> A (x > 0 ? x + 1 : exp(x))
__device__ __forceinline__ float elu1_scaled(float xs, const ActScale& A) {
  return xs > 0.f ? xs + A.a : __builtin_amdgcn_exp2f(__builtin_fmaf(xs, kLog2e * A.inv, A.log2a));
}

// completion of the LDS-DMA fragment copies of stage_frags_n (every wave waits for its own, then the barrier)
__device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
constexpr int kRegionBytes = 64 * 1024;

// ------------------------------------------------------------------------------------------------ the kernel
// A layer call x <- layer(x, src) is
//     kv phase:      for each slice of src: K, V projections, KV += K^T V, ksum += sum K   (weights k, v in LDS)
//     update phase:  for each slice of x:   q, attention with KV, merge, LN, MLP, LN, residual, stored at once
// with the slices re-read from the windows (L2-resident: a match is 25 KB).  The four calls of a match - self on image
// 1, self on image 0, cross 0 <- 1, cross 1 <- updated 0 - run in place on the output windows.
struct KvState {
  half8 ah[4], al[4];     // KV rows d as operand A of msg^T = KV^T . Q^T, per k-step over d (32 registers; the
};                        // normaliser's operand Kd is rebuilt from ksum in LDS where it is used)

// Kd fragment of k-step s: row = head r (< 8), element j = sum_s K of feature d = 16 s + 8 (j>>2) + 4 h + (j&3) if in head r
__device__ __forceinline__ void kd_fragment(const float* ksum_lds, int s, int r, int h, half8& dh, half8& dl, float& amax,
                                            const ActScale& A) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
    // (ksum_lds is in the operand scale; the sum's own scale follows the activation scale - kSumScale at 2^8 - so that
    // a match whose activations forced a smaller scale gets room for its sums as well)
    const float val = (r == (d >> 3)) ? ksum_lds[d] * (kSumScale / 256.f) : 0.f;
    amax = fmaxf(amax, val);                  // sum_s (elu(k) + 1) of a head feature must stay below 2047 at scale 2^8
    const _Float16 hh = (_Float16)val;
    dh[j] = hh;
    dl[j] = (_Float16)(val - (float)hh);
  }
}

// [OT x 32 tokens] = W . X^T for one slice: rows[i] = the i-th 32-feature row tile of the source
template <int OT, int KS, int KSW>
__device__ __forceinline__ void gemm_T1(f32x16 (&out)[OT], const f32x16* const (&rows)[KS / 2], const half8* wf,
                                        const half8* wfl, int lane, float& amax) {
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) zero(out[ot]);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    half8 bh, bl;
    split8(*rows[s >> 1], s & 1, bh, bl, amax);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const int fi = (ot * KSW + s) * 64 + lane;
      mma3(out[ot], wf[fi], wfl[fi], bh, bl);
    }
  }
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) rescale(out[ot], 1.0f / kWgtScale);
}

__device__ __forceinline__ void layer_norm_T1(f32x16 (&y)[2], const float* __restrict__ gamma, const float* __restrict__ beta, int h,
                                              const ActScale& A) {
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int g = 0; g < 16; ++g) s += y[rt][g];
  const float mean = swap_halves_add(s) * (1.0f / 64.0f);
  float v = 0.f;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int g = 0; g < 16; ++g) { const float d = y[rt][g] - mean; v += d * d; }
  const float rstd = A.a / sqrtf(swap_halves_add(v) * (1.0f / 64.0f) + 1e-5f * A.a * A.a);   // scaled eps; x the output's scale
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 ga = *reinterpret_cast<const float4*>(gamma + 32 * rt + 8 * q + 4 * h);
      const float4 be = *reinterpret_cast<const float4*>(beta + 32 * rt + 8 * q + 4 * h);
      const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) y[rt][4 * q + e] = __builtin_fmaf((y[rt][4 * q + e] - mean) * rstd, gg[e], bb[e] * A.a);
    }
}

// slice ct of a window [WW, 64] (token-major) <-> T layout: lane (token r, half h) holds features 32 rt + 8 q + 4 h + 0..3
template <int WW>
__device__ __forceinline__ void load_slice(f32x16 (&x)[2], const float* win, int ct, int lane, const ActScale& A) {
  const int tok = 32 * ct + (lane & 31), h = lane >> 5;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tok < WW) v = *reinterpret_cast<const float4*>(win + tok * 64 + 32 * rt + 8 * q + 4 * h);
      x[rt][4 * q] = v.x * A.a; x[rt][4 * q + 1] = v.y * A.a;      // into the operand scale
      x[rt][4 * q + 2] = v.z * A.a; x[rt][4 * q + 3] = v.w * A.a;
    }
}
template <int WW>
__device__ __forceinline__ void store_slice(const f32x16 (&x)[2], float* win, int ct, int lane, const ActScale& A) {
  const int tok = 32 * ct + (lane & 31), h = lane >> 5;
  if (tok >= WW) return;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(win + tok * 64 + 32 * rt + 8 * q + 4 * h) =
          make_float4(x[rt][4 * q] * A.inv, x[rt][4 * q + 1] * A.inv, x[rt][4 * q + 2] * A.inv, x[rt][4 * q + 3] * A.inv);
}

// kv phase: region B holds [K hi 8 | V hi 8 | K lo 8 | V lo 8] fragments
template <int WW>
__device__ __forceinline__ void kv_phase(const float* src, const half8* lb, float* ksum_lds, KvState& st, int lane, float& amax,
                                         const ActScale& A) {
  constexpr int NCT = (WW + 31) / 32;
  const int r = lane & 31, h = lane >> 5;
  const half8 *wk = lb, *wv_ = lb + 8 * 64, *wkl = lb + 16 * 64, *wvl = lb + 24 * 64;
  f32x16 kv[2];
  zero(kv[0]); zero(kv[1]);
  float ks[2] = {0.f, 0.f};
#pragma unroll 1
  for (int ct = 0; ct < NCT; ++ct) {
    // (the weight fragments sit at the same LDS addresses for every slice: without this the compiler hoists all of
    // their loads out of the slice loop and spills them)
    asm volatile("" : "+v"(lane));
    f32x16 x[2];
    load_slice<WW>(x, src, ct, lane, A);
    // K, V [32 tokens x 64] = S . W^T: the T-layout registers of S as operand A, N layout out (tokens in registers,
    // features on lanes)
    f32x16 k[2], v[2];
    zero(k[0]); zero(k[1]); zero(v[0]); zero(v[1]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      half8 ah, al;
      split8(x[s >> 1], s & 1, ah, al, amax);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int fi = (ot * 4 + s) * 64 + lane;
        mma3(k[ot], ah, al, wk[fi], wkl[fi]);
        mma3(v[ot], ah, al, wv_[fi], wvl[fi]);
      }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const bool tok_ok = 32 * ct + (g & 3) + 8 * (g >> 2) + 4 * h < WW;      // padded tokens stay out of the sums
        k[ot][g] = tok_ok ? elu1_scaled(k[ot][g] * (1.0f / kWgtScale), A) : 0.f;
        v[ot][g] = v[ot][g] * (1.0f / kWgtScale) * (1.0f / (float)WW);          // values / S (attentions.py:41-42)
        ks[ot] += k[ot][g];
      }
    // KV += K^T . V over this slice's tokens: only the two diagonal 32 x 32 tiles hold head blocks
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8 ah, al, bh, bl;
        split8(k[dt], s, ah, al, amax);
        split8(v[dt], s, bh, bl, amax);
        mma3(kv[dt], ah, al, bh, bl);
      }
  }
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) {
    const float t = swap_halves_add(ks[ot]);
    if (h == 0) ksum_lds[32 * ot + r] = t;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 16; ++g)            // keep d / 8 == v / 8 (rows d = (g&3) + 8 (g>>2) + 4 h, column v = r)
      kv[dt][g] = ((g >> 2) != (r >> 3)) ? 0.f : kv[dt][g] * A.inv;     // back to the operand scale
#pragma unroll
  for (int s = 0; s < 4; ++s) split8(kv[s >> 1], s & 1, st.ah[s], st.al[s], amax);
}

// update phase: region A holds [Q 8 | M 8 | W2 16 hi, then the same lo], region B [W1 32 hi | 32 lo]
template <int WW>
__device__ __forceinline__ void update_phase(const float* xin, float* xout, bool store, const KvState& st,
                                             const float* ksum_lds, const half8* la, const half8* lb,
                                             const float* __restrict__ ln, int lane, float& amax, const ActScale& A) {
  constexpr int NCT = (WW + 31) / 32;
  const int r = lane & 31, h = lane >> 5;
  const half8 *lal = la + 32 * 64, *lbl = lb + 32 * 64;
#pragma unroll 1
  for (int ct = 0; ct < NCT; ++ct) {
    asm volatile("" : "+v"(lane));          // see kv_phase
    f32x16 x[2];
    load_slice<WW>(x, xin, ct, lane, A);
    f32x16 q[2];
    {
      const f32x16* const rows[2] = {&x[0], &x[1]};
      gemm_T1<2, 4, 4>(q, rows, la, lal, lane, amax);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 16; ++g) q[rt][g] = elu1_scaled(q[rt][g], A);
    // msg^T = KV^T . Q^T and den^T = Kd . Q^T (both sum over d, the row index of KV and of Q^T)
    f32x16 msg[2], den;
    zero(msg[0]); zero(msg[1]); zero(den);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      half8 bh, bl, dh, dl;
      split8(q[s >> 1], s & 1, bh, bl, amax);
      kd_fragment(ksum_lds, s, r, h, dh, dl, amax, A);
      mma3(msg[s >> 1], st.ah[s], st.al[s], bh, bl);     // KV[dt] only reaches output rows v in tile dt
      mma3(den, dh, dl, bh, bl);
    }
    rescale(msg[0], A.inv);
    rescale(msg[1], A.inv);
    {
      // Z[token][head] = 1 / (den + eps): den rows 0..3 sit in registers 0..3 of half 0, rows 4..7 in half 1
      float z[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dsc = A.inv * A.inv * (256.f / kSumScale);        // 1 / (sum scale x activation scale)
        const float mine = den[e] * dsc, theirs = other_half(den[e]) * dsc;
        z[e] = h ? theirs : mine;             // heads 0..3
        z[4 + e] = h ? mine : theirs;         // heads 4..7
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (float)WW / (z[e] + 1e-6f);     // ... * S (attentions.py:46)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 16; ++g) msg[rt][g] *= z[4 * rt + (g >> 2)];
    }
    f32x16 m1[2];
    {
      const f32x16* const rows[2] = {&msg[0], &msg[1]};
      gemm_T1<2, 4, 4>(m1, rows, la + 8 * 64, lal + 8 * 64, lane, amax);
    }
    layer_norm_T1(m1, ln, ln + 64, h, A);
    // MLP in two halves of the hidden layer (64 of its 128 features at a time: 32 registers instead of 64):
    // m2 = sum over halves of W2[:, half] . relu(W1[half, :] . [x | m1])
    f32x16 m2[2];
    zero(m2[0]); zero(m2[1]);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      f32x16 hid[2];
      {
        const f32x16* const rows[4] = {&x[0], &x[1], &m1[0], &m1[1]};
        gemm_T1<2, 8, 8>(hid, rows, lb + (2 * hf) * 8 * 64, lbl + (2 * hf) * 8 * 64, lane, amax);
      }
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 16; ++g) hid[ot][g] = fmaxf(hid[ot][g], 0.f);
#pragma unroll
      for (int s = 0; s < 4; ++s) {             // k-steps 4 hf .. 4 hf + 3 of the second MLP matrix
        half8 bh, bl;
        split8(hid[s >> 1], s & 1, bh, bl, amax);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const int fi = (ot * 8 + 4 * hf + s) * 64 + lane;
          mma3(m2[ot], la[16 * 64 + fi], lal[16 * 64 + fi], bh, bl);
        }
      }
    }
    rescale(m2[0], 1.0f / kWgtScale);
    rescale(m2[1], 1.0f / kWgtScale);
    layer_norm_T1(m2, ln + 128, ln + 192, h, A);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 16; ++g) x[rt][g] += m2[rt][g];
    if (store) store_slice<WW>(x, xout, ct, lane, A);
    {
      // NaN never wins v_max3 / fmaxf, so `amax` alone misses it: the sum of the slice about to be stored is NaN as soon
      // as any of its values is NaN or Inf (a NaN anywhere in a window reaches every token of the match through the
      // attention sums), and s - s != 0 then
      float chk = 0.f;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int k = 0; k < 16; ++k) chk += x[rt][k];
      if (!(chk - chk == 0.f)) amax = INFINITY;
    }
  }
}

// nfrags 1 KiB fragment blocks global -> LDS by LDS-DMA, spread over the NW waves of the workgroup
template <int NW>
__device__ __forceinline__ void stage_frags_n(char* dst, const half8* src, int nfrags, int wv, int lane) {
  for (int f = wv; f < nfrags; f += NW)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
}

template <int WW, int NW>
__global__ __launch_bounds__(NW * 64) void k_fine_tf(const float* win0, const float* win1, int m_max,
                                                      const int32_t* __restrict__ d_count, const half8* __restrict__ wpack,
                                                      const float* __restrict__ lnp, float* out0, float* out1,
                                                      const int32_t* __restrict__ pack_status, int32_t* d_status,
                                                      int start_e2, int32_t* d_lowered) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // region A (64 KiB), region B (64 KiB)
  __shared__ float ksum[NW][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if ((int)blockIdx.x * NW >= M) return;                          // uniform: nothing left for this workgroup
  // one wave per match; a wave beyond the last match works on the last one again and stores nothing (the calls run
  // in place: a second writer would feed the owner's later calls with already updated slices)
  const int m = min((int)blockIdx.x * NW + wv, M - 1);
  const bool store = (int)blockIdx.x * NW + wv < M;
  const long off = (long)m * WW * 64;
  char* const ra = lds;
  char* const rb = lds + kRegionBytes;
  const half8* const la = reinterpret_cast<const half8*>(ra);
  const half8* const lb = reinterpret_cast<const half8*>(rb);
  // the four layer calls of a match: x <- layer(x, src), in place on the output windows after the first touch
  const float* xin[4] = {win1 + off, win0 + off, out0 + off, out1 + off};
  float* xout[4] = {out1 + off, out0 + off, out0 + off, out1 + off};
  const float* src[4] = {win1 + off, win0 + off, out1 + off, out0 + off};
  float amax = 0.f;                     // largest magnitude that went into a float16 operand (operand scale)
  // A match whose operands leave float16 at the current activation scale starts again from its input windows with a
  // 16x smaller one (the calls read win0 / win1 first and run in place on the outputs afterwards, so a restart is
  // clean).  The waves of a workgroup share the staged weights and their barriers: all of them repeat the pass when any
  // of them has to, the ones that were inside the range with their own scale (and the same result).
  // (the first attempt's scale: 2^8 unless the caller knows better - fm_fine_transformer_start: a module that saw its
  // matches lower the scale in the previous call starts there, and the repeated passes are gone)
  int e2 = start_e2;                    // (scalar registers: the exponent and the three floats made from it)
  // (every pass in which some wave lowers its scale is followed by another: at most 3 lowerings per wave)
#pragma unroll 1
  for (;;) {
    ActScale A;
    e2 = __builtin_amdgcn_readfirstlane(e2);
    A.a = __builtin_bit_cast(float, (127 + e2) << 23);
    A.inv = __builtin_bit_cast(float, (127 - e2) << 23);
    A.log2a = e2 == 8 ? 8.f : (e2 == 4 ? 4.f : (e2 == 0 ? 0.f : -4.f));
    amax = 0.f;
    bool again = false;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      const half8* wl = wpack + (c >> 1) * kTfLayerHalf8;
      const float* ln = lnp + (c >> 1) * kTfLayerFloats;
      __syncthreads();                    // every wave is past the previous call's use of both regions (and its stores)
      if ((c & 1) == 0) {                 // a new layer: q, merge, second MLP matrix -> region A
        stage_frags_n<NW>(ra, wl + kTfFragQ * 64, 8, wv, lane);
        stage_frags_n<NW>(ra + 8 * 1024, wl + kTfFragM * 64, 8, wv, lane);
        stage_frags_n<NW>(ra + 16 * 1024, wl + kTfFrag2 * 64, 16, wv, lane);
        stage_frags_n<NW>(ra + 32 * 1024, wl + (kTfFrags + kTfFragQ) * 64, 8, wv, lane);
        stage_frags_n<NW>(ra + 40 * 1024, wl + (kTfFrags + kTfFragM) * 64, 8, wv, lane);
        stage_frags_n<NW>(ra + 48 * 1024, wl + (kTfFrags + kTfFrag2) * 64, 16, wv, lane);
      }
      stage_frags_n<NW>(rb, wl + kTfFragK * 64, 16, wv, lane);                          // K, V hi
      stage_frags_n<NW>(rb + 16 * 1024, wl + (kTfFrags + kTfFragK) * 64, 16, wv, lane);    // K, V lo
      stage_wait();
      KvState st;
      kv_phase<WW>(src[c], lb, ksum[wv], st, lane, amax, A);
      __syncthreads();                    // every wave has read K, V: the first MLP matrix takes region B
      stage_frags_n<NW>(rb, wl + kTfFrag1 * 64, 32, wv, lane);
      stage_frags_n<NW>(rb + 32 * 1024, wl + (kTfFrags + kTfFrag1) * 64, 32, wv, lane);
      stage_wait();
      update_phase<WW>(xin[c], xout[c], store, st, ksum[wv], la, lb, ln, lane, amax, A);
      // checked after the first call and after the last: a match that needs a smaller scale usually shows it in the
      // first call, and then costs a quarter of a pass more, not a whole one (a check after every call costs the matches
      // that need none 7 %)
      if (c == 0 || c == 3) {
        // (a wave that is over the range at the smallest scale cannot be helped: it finishes the pass and the call
        // reports FM_DEV_RANGE; the pass is repeated only while somebody can still lower its scale)
        const bool lower = __any(!(amax <= 65504.f)) && e2 > -4;       // wave-uniform
        again = __syncthreads_or(lower ? 1 : 0) != 0;
        if (again) {
          if (lower) e2 -= 4;
          break;
        }
      }
    }
    if (!again) break;
  }
  // an operand left the float16 range (or is not finite), or a packed weight did (|w| >= 16): the results of this
  // match are not trustworthy - report it instead of clamping silently
  if (d_status && __any(!(amax <= 65504.f)) && lane == 0) atomicOr(d_status, (int)FM_DEV_RANGE);
  // how far below the starting scale this match ended (0, 4, 8, 12): only waves that lowered report (none in the steady
  // state of a caller that feeds the value back)
  if (d_lowered && e2 < start_e2 && lane == 0) atomicMax(d_lowered, start_e2 - e2);
  if (d_status && blockIdx.x == 0 && threadIdx.x == 0 && *pack_status) atomicOr(d_status, (int)FM_DEV_RANGE);
}

// One weight matrix W [OUT x IN] (row-major, nn.Linear.weight) -> operand fragments [out tile][k-step][lane] x 8 halves,
// hi plane at dst, lo plane at dst + kTfFrags*64: element j of lane (r, h) = W[32 ot + r][16 s + 8 (j>>2) + 4 h + (j&3)]
__global__ __launch_bounds__(256) void k_tf_pack(const float* __restrict__ w, int out_f, int in_f, half8* __restrict__ dst,
                                                 int32_t* __restrict__ pack_status) {
  const int ks = in_f / 16;
  const int idx = blockIdx.x * 256 + threadIdx.x;      // (ot, s, lane)
  if (idx >= out_f / 32 * ks * 64) return;
  const int lane = idx & 63, s = (idx >> 6) % ks, ot = (idx >> 6) / ks;
  const int r = lane & 31, h = lane >> 5;
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = w[(long)(32 * ot + r) * in_f + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)] * kWgtScale;
    if (!(fabsf(x) <= 65504.f)) *pack_status = 1;        // |w| >= 16 (or not finite): beyond the fixed weight scale
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
__global__ void k_tf_clear_status(int32_t* st) { st[threadIdx.x] = 0; }

}  // namespace fm

using namespace fm;

// fragments of both layers, LayerNorm terms of both layers, then 16 bytes of pack status (word 0 != 0: a weight lies
// outside the fixed weight scale)
constexpr size_t kTfPackStatusOff = 2 * ((size_t)kTfLayerHalf8 * 16 + kTfLayerFloats * 4);
extern "C" size_t fm_fine_tf_packed_bytes(void) { return kTfPackStatusOff + 16; }

// layer_weights[l] for l = 0 ('self'), 1 ('cross'): pointers to q_proj, k_proj, v_proj, merge [64,64], mlp.0 [128,128],
// mlp.2 [64,128], norm1.weight, norm1.bias, norm2.weight, norm2.bias [64]  (10 device pointers per layer)
extern "C" int fm_fine_tf_pack_weights(const float* const* layer0, const float* const* layer1, void* packed, void* stream) {
  if (!layer0 || !layer1 || !packed) return FM_E_NULL;
  for (int i = 0; i < 10; ++i)
    if (!layer0[i] || !layer1[i]) return FM_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  half8* frag = (half8*)packed;
  float* ln = (float*)((char*)packed + 2 * (size_t)kTfLayerHalf8 * 16);
  int32_t* pstat = (int32_t*)((char*)packed + kTfPackStatusOff);
  hipLaunchKernelGGL(k_tf_clear_status, dim3(1), dim3(4), 0, st, pstat);
  const int base[6] = {kTfFragQ, kTfFragK, kTfFragV, kTfFragM, kTfFrag1, kTfFrag2};
  const int outf[6] = {64, 64, 64, 64, 128, 64}, inf[6] = {64, 64, 64, 64, 128, 128};
  for (int l = 0; l < 2; ++l) {
    const float* const* w = l ? layer1 : layer0;
    for (int i = 0; i < 6; ++i) {
      const int n = outf[i] / 32 * (inf[i] / 16) * 64;
      hipLaunchKernelGGL(k_tf_pack, dim3((n + 255) / 256), dim3(256), 0, st, w[i], outf[i], inf[i],
                         frag + (size_t)l * kTfLayerHalf8 + base[i] * 64, pstat);
    }
    for (int i = 0; i < 4; ++i)        // gamma and beta as they are (k_fine_tf applies the wave's activation scale)
      hipLaunchKernelGGL(k_tf_scale_copy, dim3(1), dim3(64), 0, st, w[6 + i], ln + l * kTfLayerFloats + 64 * i, 1.0f);
  }
  return (int)hipGetLastError();
}

extern "C" int fm_fine_transformer_status(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW,
                                          int Cf, const void* packed, float* out0, float* out1, int32_t* d_status,
                                          void* stream) {
  return fm_fine_transformer_start(win0, win1, m_max, d_count, WW, Cf, packed, out0, out1, d_status, (int)kLog2ActScale, nullptr,
                                   stream);
}

extern "C" int fm_fine_transformer_start(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW,
                                         int Cf, const void* packed, float* out0, float* out1, int32_t* d_status,
                                         int start_log2_scale, int32_t* d_lowered, void* stream) {
  if (start_log2_scale != 8 && start_log2_scale != 4 && start_log2_scale != 0 && start_log2_scale != -4) return FM_E_UNSUPPORTED;
  if (m_max == 0) return FM_OK;
  if (!win0 || !win1 || !packed || !out0 || !out1) return FM_E_NULL;
  if (m_max < 0) return FM_E_SHAPE;
  if (Cf != 64 || (WW != 25 && WW != 49)) return FM_E_UNSUPPORTED;
  const half8* frag = (const half8*)packed;
  const float* ln = (const float*)((const char*)packed + 2 * (size_t)kTfLayerHalf8 * 16);
  const int32_t* pstat = (const int32_t*)((const char*)packed + kTfPackStatusOff);
  hipStream_t st = (hipStream_t)stream;
  constexpr int NW = 8;               // matches (waves) per workgroup: two waves per SIMD share the staged weights
  const int blocks = (m_max + NW - 1) / NW;
  const int smem = 2 * kRegionBytes;
  static unsigned long long set49 = 0, set25 = 0;
  if (WW == 49) {
    hipError_t e = ensure_dynamic_lds(&k_fine_tf<49, NW>, smem, &set49);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_fine_tf<49, NW>), dim3(blocks), dim3(NW * 64), smem, st, win0, win1, m_max, d_count, frag, ln, out0, out1, pstat, d_status, start_log2_scale, d_lowered);
  } else {
    hipError_t e = ensure_dynamic_lds(&k_fine_tf<25, NW>, smem, &set25);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_fine_tf<25, NW>), dim3(blocks), dim3(NW * 64), smem, st, win0, win1, m_max, d_count, frag, ln, out0, out1, pstat, d_status, start_log2_scale, d_lowered);
  }
  return (int)hipGetLastError();
}

extern "C" int fm_fine_transformer(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                                   const void* packed, float* out0, float* out1, void* stream) {
  return fm_fine_transformer_status(win0, win1, m_max, d_count, WW, Cf, packed, out0, out1, nullptr, stream);
}
