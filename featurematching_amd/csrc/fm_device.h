// Device-side helpers shared by the coarse kernels.
#pragma once
#include "fm_internal.h"

namespace fm {

// Order-preserving uint code of a float (larger float <=> larger code; every code of a real number is > 0,
// so zero-initialised memory acts as "nothing recorded yet" under atomicMax).  max is exact and order
// independent, so the maxima published this way are deterministic without a reduction kernel.
__device__ __forceinline__ unsigned ord_encode(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_decode(unsigned c) {      // c == 0 (nothing recorded) -> 0
  if (c == 0u) return 0.f;
  return __uint_as_float((c & 0x80000000u) ? (c & 0x7fffffffu) : ~c);
}

// float16 / bfloat16 bit patterns -> float32 (exact)
__device__ __forceinline__ float half_bits_to_float(unsigned h, int dtype) {       // low 16 bits of h
  if (dtype == FM_BF16) return __uint_as_float(h << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}
__device__ __forceinline__ float4 half4_to_float4(uint2 p, int dtype) {
  return make_float4(half_bits_to_float(p.x & 0xffffu, dtype), half_bits_to_float(p.x >> 16, dtype),
                     half_bits_to_float(p.y & 0xffffu, dtype), half_bits_to_float(p.y >> 16, dtype));
}

// Exact power-of-two scale of an image's float16 hi / lo planes: 2^k with amax * 2^k in [2^13, 2^14) (1 for an all-zero
// or bad image).  The matrix cores flush float16 SUBNORMAL inputs, so without it the lo half of every value below
// 2^-3 - |lo| ~ 2^-12 |x| < 2^-14 - would be lost (such elements would carry 11 instead of 22 bits).
__device__ __forceinline__ float f16_plane_scale(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
  int e;
  frexpf(amax, &e);                       // amax = f * 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 14 - e);
}

// Error of the int8 screening product against the exact one (raw dot-product units).  k_prep_split quantises each
// IMAGE with ONE step: q = clamp(rint(a / sigma), -127, 127).  With a = sigma q + da, |da_k| <= sigma / 2 + e_k where
// e_k = max(|a_k| - 127 sigma, 0) is what the clamp cut off (zero unless the step, estimated from a sample of the
// image's rows, turned out too small for an outlier), and likewise for b:
//   a.b - sigma_a sigma_b (q_a.q_b) = da.b + (sigma_a q_a).db
//   |da.b|            <= (sigma_a / 2) ||b||_1 + ||e_a||_1 max|b_k|,   max|b_k| <= 127 sigma_b + ||e_b||_1
//   |(sigma_a q_a).db| <= (sigma_b / 2) (||a||_1 + C sigma_a / 2) + 127 sigma_a ||e_b||_1
// clip_a / clip_b = upper bounds of ||e_a||_1 / ||e_b||_1 (the images' largest: a clipped element widens every margin
// of its image, which keeps the margin separable into a row and a column part).  The factor 1.0001 covers the float
// roundings of x / sigma, of sigma_a sigma_b and of the scaled accumulator (each ~1e-7 relative against terms of the
// same form).  Pass block / image maxima of the L1 norms to bound a whole row, unit or pair.
__device__ __forceinline__ float q8_margin_raw(float sig_a, float l1_a, float clip_a, float sig_b, float l1_b, float clip_b,
                                               float cpad) {
#pragma clang fp contract(off)
  const float clip = clip_a * (127.f * sig_b + clip_b) + 127.f * sig_a * clip_b;
  return (0.5f * sig_a * l1_b + 0.5f * sig_b * (l1_a + 0.5f * cpad * sig_a) + clip) * 1.0001f;
}

// Row / column / unit maxima of the integer screening product q_a.q_b travel as biased unsigned codes: larger
// product <=> larger code, every code of a real product is > 0 (|q.q| <= 127^2 * 256 < 2^23), so zero-initialised
// memory is "nothing recorded yet" under atomicMax - exact and order independent.
constexpr int kQBias = 0x40000000;
constexpr int kQMasked = -kQBias;          // accumulator value of padded rows / columns: code 0
__device__ __forceinline__ unsigned q_encode(int q) { return (unsigned)(q + kQBias); }
__device__ __forceinline__ float q_decode(unsigned c) { return (float)((int)c - kQBias); }   // c == 0 -> -2^30 (never the maximum of a real row)

// -stabiliser * log2(e) of a row / column whose largest screening product is `raw`, E = its margin in raw units
// (network/utils/coarse_matching_new.py:64-68: sim = raw / (C T)).  The stabiliser is the LOWER bound
// m^ = (max~ - E) / (C T) of the true maximum: every s - m^ <= 2E/(C T) (no overflow in exp2) and
// conf > thr => softmax > thr => s - m^ > ln thr, the screening test of the sum kernels.
__device__ __forceinline__ float neg_stabiliser_log2(float raw, float margin_raw, float inv_ct) {
#pragma clang fp contract(off)
  const float mhat = (raw - margin_raw) * inv_ct - 1e-6f;
  return -mhat * kLog2e;
}

// log2-domain bound of k * |screening product - exact product| from a raw margin: turns a screening product into
// an upper bound of the exact similarity.
__device__ __forceinline__ float margin_log2(float margin_raw, float inv_ct) {
#pragma clang fp contract(off)
  return margin_raw * inv_ct * kLog2e + 1e-3f;
}

}  // namespace fm
