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

// Error of the single-plane (float16 hi x hi) product against the exact one, in similarity units:
//   |fl16(a).fl16(b) - a.b| <= 2^-10 (1+2^-12) |a||b| for normal halves, + 2^-25 per operand below the half
//   normal range  =>  E = (2^-10 * 1.01 * |a| * max|b| + 2^-24 sqrt(C) (|a| + max|b|)) / (C T)  (+ 1e-6 slack).
// `nrm` = |a_i| (this row / column), `om` = largest descriptor norm of the OTHER image.
__device__ __forceinline__ float f16_product_margin(float nrm, float om, float inv_ct, float sqrt_c) {
#pragma clang fp contract(off)
  return (9.8633e-4f * nrm * om + 5.9605e-8f * sqrt_c * (nrm + om)) * inv_ct + 1e-6f;
}

// -stabiliser * log2(e) of a row / column whose largest f16 product is `raw` (network/utils/
// coarse_matching_new.py:64-68: sim = raw / (C T)).  The stabiliser is the LOWER bound m^ = max~ - E of the
// true maximum: every s - m^ <= 2E (no overflow in exp2) and conf > thr => softmax > thr => s - m^ > ln thr,
// the screening test of the sum kernels.
__device__ __forceinline__ float neg_stabiliser_log2(float raw, float nrm, float om, float inv_ct, float sqrt_c) {
#pragma clang fp contract(off)
  const float e = f16_product_margin(nrm, om, inv_ct, sqrt_c);
  const float mhat = raw * inv_ct - e;
  return -mhat * kLog2e;
}

// log2-domain bound of k * |f16 product - exact product| over the whole pair (own / om = largest norms of the
// two images): turns an f16 product into an upper bound of the exact similarity.
__device__ __forceinline__ float pair_margin_log2(float own, float om, float inv_ct, float sqrt_c) {
#pragma clang fp contract(off)
  return f16_product_margin(own, om, inv_ct, sqrt_c) * kLog2e + 1e-3f;
}

}  // namespace fm
