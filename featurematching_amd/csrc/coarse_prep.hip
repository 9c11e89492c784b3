// Coarse stage, kernels around the correlation sweeps:
//   k_prep_split  : float32 descriptors -> two float16 planes (hi, lo = x - hi) + row norms
//   k_reduce      : partial maxima of pass A -> per-row / per-column stabilisers;
//                   partial sums of pass B -> softmax denominators
//
// Reference arithmetic being reproduced: network/utils/coarse_matching_new.py:64-68
// (sim = (f0/sqrt(C)) . (f1/sqrt(C)) / T, softmax over dim 1 and dim 2).  The float16
// pair (hi, lo) carries 22 mantissa bits, so hi*hi + hi*lo + lo*hi on the f16 matrix
// cores reproduces the float32 product to ~2^-22 relative while running at the f16
// MFMA rate; the single-plane product (hi*hi) is only used to bound the row/column
// maxima, with the rigorous margin computed in k_reduce.
#include "fm_internal.h"

namespace fm {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// C = padded channel count of the planes (64/128/256); c_in <= C = channels of the source rows,
// the planes are zero beyond c_in (a dot product does not change under zero padding).
template <int C>
__global__ __launch_bounds__(256) void k_prep_split(const float* __restrict__ src, int rows, int rows_pad, int c_in,
                                                    _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                    float* __restrict__ norms, float* __restrict__ blockmax,
                                                    unsigned* __restrict__ flags) {
  constexpr int LPR = C / 4;        // lanes per row (one float4 each)
  constexpr int RPP = 256 / LPR;    // rows per pass of the workgroup
  constexpr int kPrepRows = C >= 128 ? 8 : 16;   // == prep_rows(C)
  const int tid = threadIdx.x;
  const int sub = tid / LPR;
  const int lir = tid % LPR;
  const long row0 = (long)blockIdx.x * kPrepRows;
  float bmax = 0.f;
  bool bad = false;
#pragma unroll
  for (int p = 0; p < kPrepRows / RPP; ++p) {
    const long prow = row0 + p * RPP + sub;
    const int b = (int)(prow / rows_pad);
    const int local = (int)(prow - (long)b * rows_pad);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (local < rows && lir * 4 < c_in)
      v = *reinterpret_cast<const float4*>(src + ((long)b * rows + local) * c_in + lir * 4);
    // NaN fails the comparison too
    bad = bad || !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) < 32768.f) ||
          v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
    half4 h, l;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
    l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
    *reinterpret_cast<half4*>(hi + prow * C + lir * 4) = h;
    *reinterpret_cast<half4*>(lo + prow * C + lir * 4) = l;
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
    const float nrm = sqrtf(ss);
    if (lir == 0) norms[prow] = nrm;
    bmax = fmaxf(bmax, nrm);
  }
  __shared__ float sm[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, m));
  if ((tid & 63) == 0) sm[tid >> 6] = bmax;
  __syncthreads();
  if (tid == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  if (__any(bad) && (tid & 63) == 0) atomicOr(flags, (unsigned)FM_DEV_RANGE);
}

hipError_t launch_prep(const float* feat, int N, int rows, int rows_pad, int c_in, int C, _Float16* hi, _Float16* lo,
                       float* norms, float* blockmax, unsigned* flags, hipStream_t st) {
  const int blocks = (int)((long)N * rows_pad / prep_rows(C));
  switch (C) {
    case 64: hipLaunchKernelGGL(k_prep_split<64>, dim3(blocks), dim3(256), 0, st, feat, rows, rows_pad, c_in, hi, lo, norms, blockmax, flags); break;
    case 128: hipLaunchKernelGGL(k_prep_split<128>, dim3(blocks), dim3(256), 0, st, feat, rows, rows_pad, c_in, hi, lo, norms, blockmax, flags); break;
    case 256: hipLaunchKernelGGL(k_prep_split<256>, dim3(blocks), dim3(256), 0, st, feat, rows, rows_pad, c_in, hi, lo, norms, blockmax, flags); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// grid (chunks, N, 2): z = 0 rows of image 0 (statistic over j), z = 1 columns (over i).
// MODE 0: part = raw dot-product maxima of pass A (float16 hi plane only) -> stabilisers.
//   Error of that product against the exact one: |fl16(a) fl16(b) - a b| <= 2^-10 (1+2^-12) |a||b|
//   for normal halves, + 2^-25 per operand below the half normal range, so
//     |max~ - max| <= E = (2^-10 * 1.01 * |a_i| * max_j|b_j| + 2^-24 sqrt(C) (|a_i| + max_j|b_j|)) / (C T).
//   The stabiliser is the LOWER bound m^ = max~ - E: then every s - m^ <= 2E (no overflow in exp)
//   and conf > thr  =>  softmax > thr  =>  s - m^ > ln(thr), which is the screening test of pass B.
//   out = -m^ * log2(e).
// MODE 1: part = partial sums of exp(s - m^) of pass B -> out = their sum (fixed order: deterministic).
// Workgroup = 16 consecutive entries x 16 part-groups: every thread folds ~nparts/16 partials (all
// loads independent and in flight together), then the 16 groups are folded through LDS in a fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void k_reduce(const float* __restrict__ rowP, const float* __restrict__ colP,
                                                const float* __restrict__ norm0, const float* __restrict__ norm1,
                                                const float* __restrict__ bmax0, const float* __restrict__ bmax1,
                                                float* __restrict__ rout, float* __restrict__ cout_, int Lp, int Sp,
                                                int rparts, int cparts, float inv_ct, float sqrt_c, int prows,
                                                const float* __restrict__ nm_r, const float* __restrict__ nm_c,
                                                float* __restrict__ nm2_r, float* __restrict__ nm2_c,
                                                int* __restrict__ cand_count, const unsigned* __restrict__ flags) {
  const int side = blockIdx.z;
  const int b = blockIdx.y;
  const int len = side ? Sp : Lp;
  if ((int)blockIdx.x * 16 >= len) return;
  const int nparts = side ? cparts : rparts;
  const float* part = (side ? colP : rowP) + (long)b * nparts * len;
  float* out = (side ? cout_ : rout) + (long)b * len;
  const int cx = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + cx;       // len is a multiple of 64

  __shared__ float fold[16][17];
  __shared__ float sm[4];
  float acc = MODE ? 0.f : -INFINITY;
#pragma unroll 4
  for (int p = pg; p < nparts; p += 16) {
    const float v = part[(long)p * len + idx];
    acc = MODE ? acc + v : fmaxf(acc, v);
  }
  fold[pg][cx] = acc;

  float om = 0.f;
  if (MODE == 0) {   // largest descriptor norm of the OTHER image (for the screening margin)
    const int other_len = side ? Lp : Sp;
    const float* obm = (side ? bmax0 : bmax1) + (long)b * (other_len / prows);
    for (int k = threadIdx.x; k < other_len / prows; k += 256) om = fmaxf(om, obm[k]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) om = fmaxf(om, __shfl_xor(om, m));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = om;
  }
  __syncthreads();
  if (pg != 0) return;
  float v = fold[0][cx];
#pragma unroll
  for (int g = 1; g < 16; ++g) v = MODE ? v + fold[g][cx] : fmaxf(v, fold[g][cx]);
  if (MODE == 1) {
    out[idx] = v;
    // log-softmax offset for the exact screening of pass C: log2 P = x*k + (nm - log2(sum))
    const long gi = (long)b * len + idx;
    (side ? nm2_c : nm2_r)[gi] = (side ? nm_c : nm_r)[gi] - __log2f(v);
    // pass B overflowed some row's candidate slots: pass C refills the lists from scratch
    if (side == 0 && (*flags & FM_INT_SCREEN_OVERFLOW)) cand_count[gi] = 0;
    return;
  }
  om = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float raw = v;
  if (!(raw > -INFINITY)) raw = 0.f;   // padded row/column: never used
  const float nrm = ((side ? norm1 : norm0) + (long)b * len)[idx];
  const float e = (9.8633e-4f * nrm * om + 5.9605e-8f * sqrt_c * (nrm + om)) * inv_ct + 1e-6f;
  const float mhat = raw * inv_ct - e;
  out[idx] = -mhat * kLog2e;
}

hipError_t launch_reduce(int mode, const CoarseWs& w, char* base, float inv_ct, hipStream_t st) {
  const int chunks = (max(w.Lp, w.Sp) + 15) / 16;
  const dim3 grid(chunks, w.N, 2);
  const float* n0 = (const float*)(base + w.norm0); const float* n1 = (const float*)(base + w.norm1);
  const float* b0 = (const float*)(base + w.bmax0); const float* b1 = (const float*)(base + w.bmax1);
  if (mode == 0)
    hipLaunchKernelGGL(k_reduce<0>, grid, dim3(256), 0, st, (const float*)(base + w.rowA),
                       (const float*)(base + w.colA), n0, n1, b0, b1, (float*)(base + w.nmr), (float*)(base + w.nmc),
                       w.Lp, w.Sp, w.splits, w.panels * 8, inv_ct, sqrtf((float)w.C), prep_rows(w.C), nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL(k_reduce<1>, grid, dim3(256), 0, st, (const float*)(base + w.rowB),
                       (const float*)(base + w.colB), n0, n1, b0, b1, (float*)(base + w.rsum), (float*)(base + w.csum),
                       w.Lp, w.Sp, w.splits, w.panels * 8, inv_ct, sqrtf((float)w.C), prep_rows(w.C),
                       (const float*)(base + w.nmr), (const float*)(base + w.nmc), (float*)(base + w.nmr2),
                       (float*)(base + w.nmc2), (int*)(base + w.cand_count), (const unsigned*)(base + w.scalars));
  return hipGetLastError();
}

}  // namespace fm
