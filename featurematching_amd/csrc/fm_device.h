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

// Error of the int8 screening product against the exact one (raw dot-product units).  With a = sigma_a q_a + da,
// |da_k| <= sigma_a / 2 (k_prep_split: q = rint(a / sigma), sigma = block max / 127) and likewise for b:
//   a.b - sigma_a sigma_b (q_a.q_b) = da.b + (a - da).db
//   |...| <= (sigma_a / 2) ||b||_1 + (sigma_b / 2) (||a||_1 + C sigma_a / 2).
// The factor 1.0001 covers the float roundings of x / sigma, of sigma_a sigma_b and of the scaled accumulator
// (each ~1e-7 relative against terms of the same form).  Pass block / image maxima of sigma and of the L1 norms to
// bound a whole row, unit or pair.
__device__ __forceinline__ float q8_margin_raw(float sig_a, float l1_a, float sig_b, float l1_b, float cpad) {
#pragma clang fp contract(off)
  return (0.5f * sig_a * l1_b + 0.5f * sig_b * (l1_a + 0.5f * cpad * sig_a)) * 1.0001f;
}

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
