// Match post-processing on the GPU: the step after the path.
//
// Squared symmetric epipolar distance of every match against the ground-truth / predicted relative pose
// (utils/metrics.py:33-57 symmetric_epipolar_distance, :60-81 compute_symmetrical_epipolar_errors) and a
// RANSAC-free inlier score per pair: matches and inliers (d < thr).
//   E = [t]x R from T_0to1 (:65-66), points normalised by their intrinsics (:41-42), homogeneous (:43-44),
//   d = (p1 . E p0)^2 (1 / ((E p0)_x^2 + (E p0)_y^2) + 1 / ((E^T p1)_x^2 + (E^T p1)_y^2))   (:47-56)
// One thread per match; the per-pair counters are integer atomics (exact, order independent).
#include "fm_internal.h"

namespace fm {

__global__ __launch_bounds__(256) void k_epipolar(const float* __restrict__ k0, const float* __restrict__ k1, int kstride,
                                                  const int64_t* __restrict__ m_bids,
                                                  const int32_t* __restrict__ d_count, int m_max, int N,
                                                  const float* __restrict__ T, const float* __restrict__ K0,
                                                  const float* __restrict__ K1, float thr, float* __restrict__ epi,
                                                  unsigned char* __restrict__ inlier, int32_t* __restrict__ per_pair) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if (m >= M) return;
  const int b = (int)m_bids[m];
  if (b < 0 || b >= N) { epi[m] = __builtin_nanf(""); if (inlier) inlier[m] = 0; return; }
  const float* t4 = T + (long)b * 16;          // row-major [4,4]
  const float tx = t4[3], ty = t4[7], tz = t4[11];
  // E = [t]x R   (numeric.cross_product_matrix(t) @ R)
  float E[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r0 = t4[c], r1 = t4[4 + c], r2 = t4[8 + c];
    E[0][c] = -tz * r1 + ty * r2;
    E[1][c] = tz * r0 - tx * r2;
    E[2][c] = -ty * r0 + tx * r1;
  }
  const float* a = K0 + (long)b * 9;
  const float* c1 = K1 + (long)b * 9;
  const float p0x = (k0[(long)m * kstride] - a[2]) / a[0], p0y = (k0[(long)m * kstride + 1] - a[5]) / a[4];
  const float p1x = (k1[(long)m * kstride] - c1[2]) / c1[0], p1y = (k1[(long)m * kstride + 1] - c1[5]) / c1[4];
  // Ep0 = p0 @ E^T, Etp1 = p1 @ E  (homogeneous coordinate 1)
  const float e0 = E[0][0] * p0x + E[0][1] * p0y + E[0][2];
  const float e1 = E[1][0] * p0x + E[1][1] * p0y + E[1][2];
  const float e2 = E[2][0] * p0x + E[2][1] * p0y + E[2][2];
  const float p1Ep0 = p1x * e0 + p1y * e1 + e2;
  const float f0 = p1x * E[0][0] + p1y * E[1][0] + E[2][0];
  const float f1 = p1x * E[0][1] + p1y * E[1][1] + E[2][1];
  const float d = p1Ep0 * p1Ep0 * (1.0f / (e0 * e0 + e1 * e1) + 1.0f / (f0 * f0 + f1 * f1));
  epi[m] = d;
  const bool in = d < thr;
  if (inlier) inlier[m] = in ? 1 : 0;
  if (per_pair) {
    atomicAdd(&per_pair[b * 2], 1);
    if (in) atomicAdd(&per_pair[b * 2 + 1], 1);
  }
}

}  // namespace fm

using namespace fm;

extern "C" int fm_epipolar_errors(const float* mkpts0, const float* mkpts1, int kpt_stride, const int64_t* m_bids,
                                  const int32_t* d_count, int m_max, int N, const float* T_0to1, const float* K0,
                                  const float* K1, float inlier_thr, float* epi_errs, unsigned char* inlier,
                                  int32_t* per_pair, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!mkpts0 || !mkpts1 || !m_bids || !T_0to1 || !K0 || !K1 || !epi_errs) return FM_E_NULL;
  if (m_max < 0 || N <= 0 || kpt_stride < 2) return FM_E_SHAPE;
  hipLaunchKernelGGL(k_epipolar, dim3((m_max + 255) / 256), dim3(256), 0, (hipStream_t)stream, mkpts0, mkpts1,
                     kpt_stride, m_bids, d_count, m_max, N, T_0to1, K0, K1, inlier_thr, epi_errs,
                     inlier, per_pair);
  return (int)hipGetLastError();
}
