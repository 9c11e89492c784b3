// Coarse stage, kernels around the correlation sweeps:
//   k_prep_split  : descriptors -> int8 screening plane (one step per image) + L1 norms, clipped mass, block maxima
//   k_prep_f16    : two float16 planes (hi, lo = x - hi) for the samples the dense sum kernel redoes
//   k_reduce_sums : lists of significant entries (k_screen) / partial sums (dense sum kernel) -> softmax denominators
//                   of every row / column (exact screening and dense conf_matrix only)
//
// Reference arithmetic being reproduced: network/utils/coarse_matching_new.py:64-68
// (sim = (f0/sqrt(C)) . (f1/sqrt(C)) / T, softmax over dim 1 and dim 2).  The float16
// pair (hi, lo) carries 22 mantissa bits, so hi*hi + hi*lo + lo*hi on the f16 matrix
// cores reproduces the float32 product to ~2^-22 relative while running at the f16
// MFMA rate (dense sum kernel).  Screening (row / column / unit maxima, which units and entries matter) runs on
// the int8 plane at twice that rate and half the bytes, with the rigorous quantisation margin of fm_device.h.
#include "fm_device.h"

namespace fm {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct PrepArgs {
  const void* src0; const void* src1; int in_dtype;      // FM_F32 / FM_F16 / FM_BF16 rows [N, rows, c_in]
  _Float16* hi0; _Float16* lo0; _Float16* hi1; _Float16* lo1;      // k_prep_f16 only
  const int* dense_cnt; int force; float* f16inv;        // k_prep_f16: which samples need planes; 1 / (scale0 scale1)
  const float4* bstat0r; const float4* bstat1r; int N;
  signed char* q0; signed char* q1;   // int8 screening planes (fragment-major for v_mfma_i32_32x32x32_i8)
  float* sigimg;                      // [N][2] the quantisation step of image 0 / image 1 of every sample
  int exact_step;                     // FM_MODE_EXACT_STEP: k_prep_amax left every block's largest |x| in bstat*.z
  float* l1_0; float* l1_1;           // L1 norm of every descriptor
  float4* bstat0; float4* bstat1;     // per 32-row block: {largest L1 norm (+inf: the block holds a bad value), largest
                                      // clipped L1 mass of a descriptor, largest |x|, 0}
  uint4* zero; int zero_vec;          // per-call counters to clear (uint4 units)
  int L, S, Lp, Sp, c_in, blocks0;    // blocks0 = workgroups that convert image 0
  float* diag;                        // diagnostic build: stamp buffer
};

__device__ __forceinline__ bool bad_value(float4 v) {   // NaN fails the comparison too
  return !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) < 32768.f) ||
         v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
}

// One dispatch prepares both images and clears the per-call counters.
// C = padded channel count of the planes (64/128/256); c_in <= C = channels of the source rows,
// the planes are zero beyond c_in (a dot product does not change under zero padding).
//   All planes are FRAGMENT-major, one workgroup per 32-row block.
//   int8 plane (the screening product of the max pass and the sparse sum kernel): q = clamp(rint(x / sigma), +-127)
//   with ONE step sigma per IMAGE: the screening product sigma_0 sigma_1 (q_i . q_j) is then ordered like the integer
//   dot product itself, so the max pass and the sparse kernel's screening stay in integer arithmetic (maxima, compares:
//   one vector instruction per accumulator register instead of convert + two scalings + two float maxima), and identical
//   descriptors get identical codes and margins wherever they sit, which keeps exact conf ties exact
//   (coarse_matching_new.py:105-106 keeps all tied entries).  The step must not depend on a grid-wide reduction (that
//   would be a kernel boundary): every workgroup derives it from the SAME kPrepSampleRows rows spread over the image,
//   sigma = kPrepHeadroom * max|x| of that sample / 127.  Whatever exceeds 127 sigma elsewhere in the image is clipped
//   and its L1 mass enters the error margin (fm_device.h): the bounds stay rigorous for any data, outliers only cost
//   margin.  Element (row, k) lives at
//     (((row/32 * KS8 + ks) * 2 + h) * 32 + row%32) * 16 + k%16   with 16-channel chunk k/16 = h*KS8 + ks,
//   the 64 lanes of a v_mfma_i32_32x32x32_i8 operand fragment are one contiguous 1 KiB block per k-step.
//   float16 planes (hi, lo = x - hi; the dense sum kernel's float32-equivalent product) are written by k_prep_f16
//   below, only for the samples that need them: element (row, k) lives at
//     (((row/32 * KSTEPS + ks) * 2 + h) * 32 + row%32) * 8 + k%8   with chunk q = k/8 = h*KSTEPS + ks.
// A workgroup that sees a non-finite / out-of-range value reports +inf as its block L1 maximum; k_sum_sparse
// turns that into FM_DEV_RANGE (the flag word itself is cleared by this kernel, so it cannot be set here).
// PLANES (FM_MODE_FLAT: the caller expects flat similarity, every sample goes to the dense sum kernel): the float16
// hi / lo planes are written here as well, from the values this kernel holds anyway - k_prep_f16's launch and its
// second pass over the descriptors are gone.  Their power-of-two scale must not wait for the image's true maximum (a
// grid-wide reduction): it is derived from the image's int8 step, 127 sigma = kPrepHeadroom * (largest |x| of the
// sampled rows) -> [2^13, 2^14), which every workgroup - and every later kernel, from sigimg - arrives at alike.  An
// element more than ~4x beyond the sample's maximum would leave float16's range: k_stab compares the image's true
// maximum (block statistics) with the scale and reports FM_DEV_STEP, the call is repeated with the exact step.
template <int C, bool PLANES>
__global__ __launch_bounds__(256) void k_prep_split(PrepArgs a) {
  constexpr int KS8 = C / 32;
  const int tid = threadIdx.x;
#ifdef FM_DIAG_CLOCK       // diagnostic build only: constant-clock stamps (10 ns) of every workgroup's phases
  unsigned long long dg[5];
  dg[0] = __builtin_amdgcn_s_memrealtime();
#define PREP_STAMP(i) dg[i] = __builtin_amdgcn_s_memrealtime();
#else
#define PREP_STAMP(i)
#endif
  // ---- clear this workgroup's slice of the per-call counters ----
  {
    const int per = (a.zero_vec + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo_ = blockIdx.x * per, hi_ = min(lo_ + per, a.zero_vec);
    for (int k = lo_ + tid; k < hi_; k += 256) a.zero[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  __shared__ float sm[8][33], sm2[8][33];
  __shared__ float wred[3][4];
  // ---------------- one 32-row block of image 0 or image 1 ----------------
  const bool img1 = (int)blockIdx.x >= a.blocks0;
  const long rb = img1 ? (int)blockIdx.x - a.blocks0 : (int)blockIdx.x;   // row block over N*Lp/32 (N*Sp/32)
  const int rows = img1 ? a.S : a.L, rows_pad = img1 ? a.Sp : a.Lp;
  signed char* const qp = img1 ? a.q1 : a.q0;
  const int b = (int)(rb * 32 / rows_pad);
  const int r = tid & 31;
  const int local = (int)(rb * 32 - (long)b * rows_pad) + r;
  const long row_off = ((long)b * rows + local) * a.c_in;          // elements
  const void* const src = img1 ? a.src1 : a.src0;

  // Full float32 rows (the common hand-over): the 32 rows of this block are one contiguous span of the source -
  // copied to LDS with fully coalesced 16-byte loads (one row per wave instruction at C = 256) and read back in
  // the (row = lane, 8-channel chunk) order the fragment-major stores need.  Row pitch + 16 bytes: the 32 lanes
  // of a half-wave then start 4 banks apart.  The loads are ISSUED here, ahead of the sample's (which mostly hit L2):
  // the block's own rows come from HBM, and behind the sample's round trip they were a second one (5.8 us from the
  // launch until both had landed).
  constexpr int PITCH = C + 4;
  constexpr int OWN = 32 * (C / 4) / 256;        // float4 per thread
  __shared__ float tile[32 * PITCH];
  const bool staged = a.in_dtype == FM_F32 && a.c_in == C;
  float4 own[OWN];
  const int first = (int)(rb * 32 - (long)b * rows_pad);            // first row of the block inside its sample
  const int valid4 = max(0, min(32, rows - first)) * (C / 4);        // float4s that exist (the sample's tail block is short)
  if (staged) {
    // (clamped, not predicated - no load behind a branch - and NOT touched before the sample's loads are out: what lies
    // beyond the sample's rows is zeroed where the tile is written)
    const float4* span = valid4 > 0 ? reinterpret_cast<const float4*>((const float*)src + ((long)b * rows + first) * C)
                                    : reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int p = 0; p < OWN; ++p) own[p] = span[valid4 > 0 ? min(p * 256 + tid, valid4 - 1) : 0];
  }

  // ---- the image's step: largest |x| over kPrepSampleRows rows spread evenly over the image (the same rows in every
  // workgroup of the image: max is order independent, so all of them arrive at the same step).  <= 8 loads per
  // thread, all in flight together and ahead of the block's own rows ----
  float amax_s = 0.f;
  const bool exact_step = a.exact_step != 0;         // (uniform)
  if (exact_step) {
    // FM_MODE_EXACT_STEP: the image's true maximum from the block maxima k_prep_amax left in bstat*.z (one load per
    // lane; a block whose own workgroup of THIS kernel has already rewritten its bstat entry holds the same .z: the
    // block's largest |x| either way); a hair of headroom so that the roundings of x / sigma cannot push the largest
    // element beyond +-127: nothing is clipped.  (Round 4: an atomicMax per block on a memset-cleared word before - the
    // memset node alone cost 4.6 us per call.)
    const int nb = rows_pad / 32;
    const float* bz = reinterpret_cast<const float*>((img1 ? a.bstat1r : a.bstat0r) + (long)b * nb) + 2;
    for (int i2 = tid; i2 < nb; i2 += 256) amax_s = fmaxf(amax_s, bz[4 * i2]);
    amax_s = amax_s * (1.0f + 1e-5f) / kPrepHeadroom;
  } else {
    const int ns = min(kPrepSampleRows, rows);
    const int vpr = a.c_in >> 2;                       // 4-channel vectors per row (c_in % 4 == 0)
    const int total = ns * vpr;                        // <= 32 * 64
    float4 sv[8];
    // (every workgroup starts at another row of the sample: 300 workgroups asking for the same lines at the same
    // moment queue up on a few L2 channels)
    const int rot = (int)((blockIdx.x * 5u) % (unsigned)ns) * vpr;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int idx = p * 256 + tid;
      sv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < total) {
        idx += rot;
        if (idx >= total) idx -= total;
        const int t = idx / vpr, v = idx - t * vpr;
        const long e = ((long)b * rows + (long)t * rows / ns) * a.c_in + v * 4;
        if (a.in_dtype == FM_F32) sv[p] = *reinterpret_cast<const float4*>((const float*)src + e);
        else sv[p] = half4_to_float4(*reinterpret_cast<const uint2*>((const unsigned short*)src + e), a.in_dtype);
      }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)
      amax_s = fmaxf(amax_s, fmaxf(fmaxf(fabsf(sv[p].x), fabsf(sv[p].y)), fmaxf(fabsf(sv[p].z), fabsf(sv[p].w))));
  }
  if (staged) {
#pragma unroll
    for (int p = 0; p < OWN; ++p) {
      const int idx = p * 256 + tid;
      *reinterpret_cast<float4*>(&tile[(idx / (C / 4)) * PITCH + (idx % (C / 4)) * 4]) =
          idx < valid4 ? own[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax_s = fmaxf(amax_s, __shfl_xor(amax_s, m));
  if ((tid & 63) == 0) wred[0][tid >> 6] = amax_s;
  PREP_STAMP(1)
  __syncthreads();
  PREP_STAMP(2)
  amax_s = fmaxf(fmaxf(wred[0][0], wred[0][1]), fmaxf(wred[0][2], wred[0][3]));
  // (a non-finite sample: some block reports the bad value below and the call fails; the step is irrelevant then)
  const float sigma = amax_s < INFINITY ? amax_s * (kPrepHeadroom / 127.0f) : 1.0f;
  const float inv_sigma = sigma > 0.f ? 1.0f / sigma : 0.f;
  const float clip_at = 127.f * sigma;
  if (tid == 0 && rb * 32 == (long)b * rows_pad) a.sigimg[b * 2 + (img1 ? 1 : 0)] = sigma;
  const float sc16 = PLANES ? f16_plane_scale(clip_at) : 1.0f;
  _Float16* const hi = img1 ? a.hi1 : a.hi0;
  _Float16* const lo = img1 ? a.lo1 : a.lo0;

  float s1 = 0.f, amax = 0.f, clip = 0.f;
  bool bad = false;
  constexpr int NCH16 = (C / 16 + 7) / 8;     // 16-channel chunks per thread (one 16-byte store of codes each)
#pragma unroll
  for (int n = 0; n < NCH16; ++n) {           // C/16 chunks of 16 channels, 8 per pass
    const int q16 = n * 8 + (tid >> 5);
    if (q16 >= C / 16) break;                  // (C = 64: four chunks, half of the threads have none)
    float4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (staged) {
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const float4*>(&tile[r * PITCH + q16 * 16 + 4 * p]);
    } else if (local < rows) {
      if (a.in_dtype == FM_F32) {
        const float* row = (const float*)src + row_off;
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (q16 * 16 + 4 * p < a.c_in) v[p] = *reinterpret_cast<const float4*>(row + q16 * 16 + 4 * p);
      } else {
        // half-precision rows (what a ROCm backbone under autocast hands over): 4 values per 8-byte load; every
        // float16 / bfloat16 value is exact in float32, so everything downstream sees the caller's numbers
        const unsigned short* row = (const unsigned short*)src + row_off;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          uint2 hv = make_uint2(0u, 0u);
          if (q16 * 16 + 4 * p < a.c_in) hv = *reinterpret_cast<const uint2*>(row + q16 * 16 + 4 * p);
          v[p] = half4_to_float4(hv, a.in_dtype);
        }
      }
    }
    int wq[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bad = bad || bad_value(v[p]);
      const float x[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
      int wd = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ax = fabsf(x[e]);
        s1 += ax;
        amax = fmaxf(amax, ax);
        clip += fmaxf(ax - clip_at, 0.f);
        const int c = (int)fminf(fmaxf(rintf(x[e] * inv_sigma), -127.f), 127.f);
        wd |= (c & 0xff) << (8 * e);
      }
      wq[p] = wd;
    }
    const int h = q16 / KS8, ks = q16 - h * KS8;
    const long off = (((rb * KS8 + ks) * 2 + h) * 32 + r) * 16;
    *reinterpret_cast<int4*>(qp + off) = make_int4(wq[0], wq[1], wq[2], wq[3]);
    if (PLANES) {      // the two 8-channel chunks of this 16-channel chunk, in k_prep_f16's layout and arithmetic
      constexpr int KSTEPS = C / 16;
      typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        const float x[8] = {v[2 * c8].x, v[2 * c8].y, v[2 * c8].z, v[2 * c8].w,
                            v[2 * c8 + 1].x, v[2 * c8 + 1].y, v[2 * c8 + 1].z, v[2 * c8 + 1].w};
        half8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xs = x[e] * sc16;
          hh[e] = (_Float16)xs;
          ll[e] = (_Float16)(xs - (float)hh[e]);
        }
        const int q8 = 2 * q16 + c8;
        const int h2 = q8 / KSTEPS, ks2 = q8 - h2 * KSTEPS;
        const long off2 = (((rb * KSTEPS + ks2) * 2 + h2) * 32 + r) * 8;
        *reinterpret_cast<half8*>(hi + off2) = hh;
        *reinterpret_cast<half8*>(lo + off2) = ll;
      }
    }
  }
  PREP_STAMP(3)
  // per-row L1 norm and clipped mass (a row's channels sit in 8 threads); block maxima
  sm[tid >> 5][r] = s1;
  sm2[tid >> 5][r] = clip;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
  const bool anybad = __any(bad);
  if ((tid & 63) == 0) { wred[1][tid >> 6] = amax; wred[2][tid >> 6] = anybad ? 1.f : 0.f; }
  __syncthreads();
  if (tid < 32) {
    float t = 0.f, ce = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) { t += sm[g][tid]; ce += sm2[g][tid]; }
    (img1 ? a.l1_1 : a.l1_0)[rb * 32 + tid] = t;
    float bl1 = t, bce = ce;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) { bl1 = fmaxf(bl1, __shfl_xor(bl1, m)); bce = fmaxf(bce, __shfl_xor(bce, m)); }
    if (tid == 0) {
      const bool bb = wred[2][0] + wred[2][1] + wred[2][2] + wred[2][3] > 0.f;
      (img1 ? a.bstat1 : a.bstat0)[rb] = make_float4(bb ? INFINITY : bl1, bce,
                                                     fmaxf(fmaxf(wred[1][0], wred[1][1]), fmaxf(wred[1][2], wred[1][3])), 0.f);
#ifdef FM_DIAG_CLOCK
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PREP_STAMP(4)
      for (int q2 = 0; q2 < 5; ++q2) a.diag[blockIdx.x * 8 + q2] = (float)(dg[q2] & 0xffffff);
#endif
    }
  }
}

// The float16 hi / lo planes of the samples the dense sum kernel will redo (dense_cnt[b] > 0), or of every sample
// when `force` (the exact-screening and conf_matrix sweeps read them too).  Peaked data never needs them: 9.8 MB of
// writes per 640x480 pair that k_prep_split used to do unconditionally.
// Each image of a sample is scaled by an exact power of two that brings its largest |x| into [2^13, 2^14) before the
// split: the matrix cores flush float16 SUBNORMAL inputs, so without it the lo half of every value below 2^-3 -
// |lo| ~ 2^-12 |x| < 2^-14 - would be lost (such elements would carry 11 instead of 22 bits).  f16inv[b] = 1 / (scale0
// scale1) turns the accumulator back into the dot product.
// FM_MODE_EXACT_STEP: the largest |x| of every 32-row block, left in bstat*.z (k_prep_split folds them per image).  Same
// grid as k_prep_split: one workgroup per 32-row block.  NaN never wins a maximum (k_prep_split reports it), Inf does and is reported there too.
__global__ __launch_bounds__(256) void k_prep_amax(PrepArgs a) {
  const int tid = threadIdx.x;
  const bool img1 = (int)blockIdx.x >= a.blocks0;
  const long rb = img1 ? (int)blockIdx.x - a.blocks0 : (int)blockIdx.x;
  const int rows = img1 ? a.S : a.L, rows_pad = img1 ? a.Sp : a.Lp;
  const int b = (int)(rb * 32 / rows_pad);
  const int first = (int)(rb * 32 - (long)b * rows_pad);
  const int nrows = max(0, min(32, rows - first));
  const long n4 = (long)nrows * (a.c_in >> 2);                  // 4-element vectors of this block (c_in % 4 == 0)
  const void* const src = img1 ? a.src1 : a.src0;
  const long e0 = ((long)b * rows + first) * a.c_in;
  float amax = 0.f;
  for (long v = tid; v < n4; v += 256) {
    float4 x;
    if (a.in_dtype == FM_F32) x = *reinterpret_cast<const float4*>((const float*)src + e0 + v * 4);
    else x = half4_to_float4(*reinterpret_cast<const uint2*>((const unsigned short*)src + e0 + v * 4), a.in_dtype);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
  __shared__ float wred[4];
  if ((tid & 63) == 0) wred[tid >> 6] = amax;
  __syncthreads();
  if (tid == 0)        // (every block of the padded range writes: k_prep_split reads all of them)
    (img1 ? a.bstat1 : a.bstat0)[rb] = make_float4(0.f, 0.f, fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])), 0.f);
}

template <int C>
__global__ __launch_bounds__(256) void k_prep_f16(PrepArgs a) {
  constexpr int KSTEPS = C / 16;
  constexpr int NCH = C / 8 / 8;
  const int tid = threadIdx.x;
  const bool img1 = (int)blockIdx.x >= a.blocks0;
  const long rb = img1 ? (int)blockIdx.x - a.blocks0 : (int)blockIdx.x;
  const int rows = img1 ? a.S : a.L, rows_pad = img1 ? a.Sp : a.Lp;
  const int b = (int)(rb * 32 / rows_pad);
  if (!a.force && a.dense_cnt[b] == 0) return;                 // uniform
  // (force 2 - the conf sweep alone wants the planes: a sample the screening kernel served gets the hi plane only, its
  // sweep is k_dense<C, CONF_LITE>)
  const bool want_lo = a.force != 2 || a.dense_cnt[b] > 0;
  // largest |x| of both images of this sample (block maxima of k_prep_split)
  __shared__ float wred[2][4];
  float m0 = 0.f, m1 = 0.f;
  for (int i = tid; i < a.Lp / 32; i += 256) m0 = fmaxf(m0, a.bstat0r[(long)b * (a.Lp / 32) + i].z);
  for (int i = tid; i < a.Sp / 32; i += 256) m1 = fmaxf(m1, a.bstat1r[(long)b * (a.Sp / 32) + i].z);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, m)); m1 = fmaxf(m1, __shfl_xor(m1, m)); }
  if ((tid & 63) == 0) { wred[0][tid >> 6] = m0; wred[1][tid >> 6] = m1; }
  __syncthreads();
  m0 = fmaxf(fmaxf(wred[0][0], wred[0][1]), fmaxf(wred[0][2], wred[0][3]));
  m1 = fmaxf(fmaxf(wred[1][0], wred[1][1]), fmaxf(wred[1][2], wred[1][3]));
  const float sc0 = f16_plane_scale(m0), sc1 = f16_plane_scale(m1);
  const float sc = img1 ? sc1 : sc0;
  if (!img1 && rb * 32 == (long)b * rows_pad && tid == 0) a.f16inv[b] = (1.0f / sc0) * (1.0f / sc1);
  _Float16* const hi = img1 ? a.hi1 : a.hi0;
  _Float16* const lo = img1 ? a.lo1 : a.lo0;
  const int r = tid & 31;
  const int local = (int)(rb * 32 - (long)b * rows_pad) + r;
  const long row_off = ((long)b * rows + local) * a.c_in;
  const void* const src = img1 ? a.src1 : a.src0;
#pragma unroll
  for (int n = 0; n < NCH; ++n) {
    const int q = n * 8 + (tid >> 5);
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (local < rows) {
      if (a.in_dtype == FM_F32) {
        const float* row = (const float*)src + row_off;
        if (q * 8 < a.c_in) v0 = *reinterpret_cast<const float4*>(row + q * 8);
        if (q * 8 + 4 < a.c_in) v1 = *reinterpret_cast<const float4*>(row + q * 8 + 4);
      } else {
        const unsigned short* row = (const unsigned short*)src + row_off;
        uint2 h0 = make_uint2(0u, 0u), h1 = h0;
        if (q * 8 < a.c_in) h0 = *reinterpret_cast<const uint2*>(row + q * 8);
        if (q * 8 + 4 < a.c_in) h1 = *reinterpret_cast<const uint2*>(row + q * 8 + 4);
        v0 = half4_to_float4(h0, a.in_dtype);
        v1 = half4_to_float4(h1, a.in_dtype);
      }
    }
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    half8 hh, ll;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xs = x[e] * sc;
      hh[e] = (_Float16)xs;
      ll[e] = (_Float16)(xs - (float)hh[e]);
    }
    const int h = q / KSTEPS, ks = q - h * KSTEPS;
    const long off = (((rb * KSTEPS + ks) * 2 + h) * 32 + r) * 8;
    *reinterpret_cast<half8*>(hi + off) = hh;
    if (want_lo) *reinterpret_cast<half8*>(lo + off) = ll;
  }
}

static void fill_prep_args(PrepArgs& a, const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w,
                           char* base) {
  a.src0 = feat0; a.src1 = feat1; a.in_dtype = in_dtype;
  a.hi0 = (_Float16*)(base + w.hi0); a.lo0 = (_Float16*)(base + w.lo0);
  a.hi1 = (_Float16*)(base + w.hi1); a.lo1 = (_Float16*)(base + w.lo1);
  a.q0 = (signed char*)(base + w.q0); a.q1 = (signed char*)(base + w.q1);
  a.sigimg = (float*)(base + w.sigimg);
  a.exact_step = 0;
  a.l1_0 = (float*)(base + w.l1_0); a.l1_1 = (float*)(base + w.l1_1);
  a.bstat0 = (float4*)(base + w.bstat0); a.bstat1 = (float4*)(base + w.bstat1);
  a.zero = (uint4*)(base + w.zero_begin); a.zero_vec = (int)((w.zero_end - w.zero_begin) / 16);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.c_in = c_in;
  a.blocks0 = (int)((long)w.N * w.Lp / 32);
  a.dense_cnt = (const int*)(base + w.dense_cnt); a.force = 0; a.f16inv = (float*)(base + w.f16inv);
  a.bstat0r = (const float4*)(base + w.bstat0); a.bstat1r = (const float4*)(base + w.bstat1); a.N = w.N;
  a.diag = (float*)(base + w.colB);       // (diagnostic builds run on a full-size workspace)
}

hipError_t launch_prep_f16(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                           int force, hipStream_t st) {
  PrepArgs a;
  fill_prep_args(a, feat0, feat1, in_dtype, c_in, w, base);
  a.force = force;
  const int blocks = a.blocks0 + (int)((long)w.N * w.Sp / 32);
  switch (w.C) {
    case 64: hipLaunchKernelGGL(k_prep_f16<64>, dim3(blocks), dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL(k_prep_f16<128>, dim3(blocks), dim3(256), 0, st, a); break;
    case 256: hipLaunchKernelGGL(k_prep_f16<256>, dim3(blocks), dim3(256), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_prep(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                       int exact_step, int planes, hipStream_t st) {
  PrepArgs a;
  fill_prep_args(a, feat0, feat1, in_dtype, c_in, w, base);
  const int blocks = a.blocks0 + (int)((long)w.N * w.Sp / 32);
  if (exact_step) {
    a.exact_step = 1;
    hipLaunchKernelGGL(k_prep_amax, dim3(blocks), dim3(256), 0, st, a);
  }
#define FM_PREP_CASE(CC)                                                                                  \
  case CC:                                                                                                \
    if (planes) hipLaunchKernelGGL((k_prep_split<CC, true>), dim3(blocks), dim3(256), 0, st, a);          \
    else hipLaunchKernelGGL((k_prep_split<CC, false>), dim3(blocks), dim3(256), 0, st, a);                \
    break;
  switch (w.C) {
    FM_PREP_CASE(64)
    FM_PREP_CASE(128)
    FM_PREP_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_PREP_CASE
  return hipGetLastError();
}

// Softmax denominators of EVERY row and column (only the exact screening pass and the dense conf_matrix need
// them; the common path forms the denominators of its candidates in k_select).
// grid (chunks, N, 2): z = 0 rows of image 0 (sum over j), z = 1 columns (over i).
// A sample the screening kernel served: the sum of the row's / column's list of significant entries, in index order
// (the order k_select uses: same bits).  A sample the dense sum kernel redid: its partial sums (partB) in a fixed order.
// nm2 = nm - log2(out) is the log-softmax offset.
// Workgroup = 16 consecutive entries x 16 part-groups: every thread folds ~nparts/16 partials (all
// loads independent and in flight together), then the 16 groups are folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void k_reduce_sums(const float* __restrict__ rowB, const float* __restrict__ colB,
                                                     float* __restrict__ rout, float* __restrict__ cout_, int Lp, int Sp,
                                                     int rpartsB, int cparts, int slots, float k,
                                                     const float* __restrict__ nm_r, const float* __restrict__ nm_c,
                                                     float* __restrict__ nm2_r, float* __restrict__ nm2_c,
                                                     const int* __restrict__ cand_count, const int* __restrict__ cand_j,
                                                     const float* __restrict__ cand_x, const int* __restrict__ ccand_count,
                                                     const int* __restrict__ ccand_i, const float* __restrict__ ccand_x,
                                                     int* __restrict__ cand_count_b, int* __restrict__ ccand_count_b,
                                                     const int* __restrict__ dense_cnt, const Scalars* __restrict__ scal) {
  const int side = blockIdx.z;
  const int b = blockIdx.y;
  const int len = side ? Sp : Lp;
  if ((int)blockIdx.x * 16 >= len) return;
  const bool dense = dense_cnt[b] > 0;       // the dense sum kernel redid this sample
  const int cx = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + cx;       // len is a multiple of 64
  const long gi = (long)b * len + idx;
  __shared__ float fold[16][17];
  float acc = 0.f;
  if (dense) {
    const int nparts = side ? cparts : rpartsB;
    const float* part = (side ? colB : rowB) + (long)b * nparts * len;
#pragma unroll 4
    for (int p = pg; p < nparts; p += 16) acc += part[(long)p * len + idx];
  } else if (pg == 0) {
    // list terms in index order: repeatedly take the smallest key above the last one taken (lists hold <= slots <= 64
    // entries, usually one)
    const int n = min((side ? ccand_count : cand_count)[gi], slots);
    const int* key = (side ? ccand_i : cand_j) + gi * slots;
    const float* xs = (side ? ccand_x : cand_x) + gi * slots;
    const float nm = (side ? nm_c : nm_r)[gi];
    int last = -1;
    for (int t = 0; t < n; ++t) {
      int best = 0x7fffffff, at = 0;
      for (int q = 0; q < n; ++q) { const int kq = key[q]; if (kq > last && kq < best) { best = kq; at = q; } }
      acc += __builtin_amdgcn_exp2f(__builtin_fmaf(xs[at], k, nm));
      last = best;
    }
  }
  fold[pg][cx] = acc;
  __syncthreads();
  if (pg != 0) return;
  float v = fold[0][cx];
  if (dense) {
#pragma unroll
    for (int g = 1; g < 16; ++g) v += fold[g][cx];
  }
  float* out = (side ? cout_ : rout) + (long)b * len;
  out[idx] = v;
  // log-softmax offset for the exact screening / dense conf_matrix sweeps: log2 P = x*k + (nm - log2(sum))
  (side ? nm2_c : nm2_r)[gi] = (side ? nm_c : nm_r)[gi] - __log2f(v);
  // the dense sum kernel overflowed some row's candidate slots: the exact screening sweep refills the lists of the
  // samples it redid from scratch (both listings of the candidates: per row and per column)
  if (dense && (scal->flags & FM_INT_SCREEN_OVERFLOW)) {
    if (side == 0) cand_count_b[gi] = 0;
    else ccand_count_b[gi] = 0;
  }
}

hipError_t launch_reduce(int mode, const CoarseWs& w, char* base, float inv_ct, hipStream_t st) {
  (void)mode;
  const int chunks = (max(w.Lp, w.Sp) + 15) / 16;
  const dim3 grid(chunks, w.N, 2);
  hipLaunchKernelGGL(k_reduce_sums, grid, dim3(256), 0, st, (const float*)(base + w.rowB), (const float*)(base + w.colB),
                     (float*)(base + w.rsum), (float*)(base + w.csum), w.Lp, w.Sp, w.splits, w.panels, w.slots,
                     inv_ct * kLog2e, (const float*)(base + w.nmr), (const float*)(base + w.nmc), (float*)(base + w.nmr2),
                     (float*)(base + w.nmc2), (const int*)(base + w.cand_count), (const int*)(base + w.cand_j),
                     (const float*)(base + w.cand_x), (const int*)(base + w.ccand_count), (const int*)(base + w.ccand_i),
                     (const float*)(base + w.ccand_x), (int*)(base + w.cand_count_b), (int*)(base + w.ccand_count_b),
                     (const int*)(base + w.dense_cnt), (const Scalars*)(base + w.scalars));
  return hipGetLastError();
}

}  // namespace fm
