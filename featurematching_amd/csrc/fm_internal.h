// Internal declarations shared by the translation units of libfmatch_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fmatch.h"

namespace fm {

constexpr int kPanelRows = 256;   // coarse rows (image-0 cells) one workgroup owns
// column partials each panel writes (one per wave: 8 waves x 32 rows)
constexpr int kColParts = 1;         // column partials per 256-row panel (the 8 waves' partials are folded in LDS)
constexpr int kTieCap = 1023;        // listed tie losers per image; beyond it the gathers scan the match list
constexpr int kTileCols = 64;     // coarse columns (image-1 cells) per streamed tile
constexpr int kUnitsPerSplit = 64; // 32-column units one workgroup of the dense-path kernels covers at most (64-bit live mask)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kSkipLog2 = 32.f;    // terms more than 2^32 below every stabiliser are negligible (see coarse_sum_sparse.hip)
// internal status bit (not reported): pass B's max-based screening overflowed a row's slots
constexpr unsigned FM_INT_SCREEN_OVERFLOW = 16u;
constexpr unsigned FM_INT_LOOKBACK_TIMEOUT = 64u;   // k_select: a predecessor's total never showed up (reported as FM_DEV_INTERNAL)
constexpr int kPrepSampleRows = 32;  // rows of an image every k_prep_split workgroup samples for the image's int8 step
constexpr float kPrepHeadroom = 1.5f; // step = headroom * (largest |x| of the sample) / 127: what lies beyond is clipped
                                      // and accounted for in the screening margins (fm_device.h)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// descriptor channels the kernels are instantiated for; smaller C is zero-padded by k_prep_split
inline int padded_channels(int c) { return c <= 64 ? 64 : (c <= 128 ? 128 : 256); }
inline bool valid_channels(int c) { return c >= 4 && c <= 256 && c % 4 == 0; }
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// Device workspace of the coarse stage; all offsets in bytes from the base.  The regions the common path uses come
// first (`common_total` bytes); the float16 planes, the dense sum kernel's partials and candidate set and the
// softmax denominators of every row / column follow and are needed only with FM_MODE_DENSE / FM_MODE_EXACT_SCREENING /
// a conf_matrix request.
struct CoarseWs {
  int N, L, S, C, Lp, Sp, panels, tiles, splits, slots;
  int splits0;                                // column splits of the max pass (its own grid size)
  int splits_s, units_s;                      // screening kernel: column chunks per row block and 32-column units per chunk (<= 64)
  // zeroed on every call (contiguous, starts at the base)
  size_t zero_begin, cand_count, ccand_count, cand_count_b, ccand_count_b, dense_cnt, scalars, zero_end;
                                              // cand_count / ccand_count: candidates per row / per column found by the
                                              // sparse sum kernel (the same entries, listed from both sides); *_b: by
                                              // the dense one; dense_cnt [N]: units of a sample the sparse kernel left
                                              // to the dense one (> 0: the dense kernel redoes the sample)
  size_t cell0, cell1;                        // (zeroed) match index + 1 of every image-0 / image-1 cell
  size_t ties0, ties1;                        // (zeroed) [0] = count, [1..kTieCap] = matches that lost their cell to an
                                              // exactly tied match (the cell-ordered gathers pick them up)
  size_t rowmax_u, colmax_u;                  // (zeroed) q_encode'd row / column maxima of the integer screening
                                              // product (max pass: atomicMax, exact and order independent)
  size_t blocktot;                            // (zeroed) k_select: matches per workgroup | published flag
  // per-row / per-column statistics
  size_t q0, q1;                              // int8 screening planes
  size_t sigimg;                              // [N][2] the int8 step of image 0 / image 1 of every sample
  size_t imgstat;                             // [N][8] per sample {largest L1 norm, largest clipped mass, largest |x|} of
                                              // image 0, then of image 1 (max pass: the block statistics folded once)
  size_t l1_0, l1_1;                          // L1 norm per descriptor
  size_t bstat0, bstat1;                      // float4 per 32-row block: {largest L1 norm (+inf: a bad value), largest
                                              // clipped L1 mass sum_k max(|x_k| - 127 sigma, 0), largest |x|, 0}
  size_t emarg;                               // [N] log2-domain bound of k * |screening product - exact product|
  size_t rowS, colS;                          // (round 3: partial sum-exp of the sparse sum kernel; empty since the
                                              // screening kernel hands over lists of significant entries)
  size_t nmr, nmc;                            // -stabiliser*log2e per row / column
  size_t umax;                                // unit maxima [N][Lp/32][Sp/32] of the integer screening product (as float)
  size_t cand_j, cand_x;                      // k_screen: every significant entry of a row (column, exact dot product)
  size_t ccand_i, ccand_x;                    // ... of a column (row, exact dot product)
  size_t thr_r, thr_c;                        // k_thresh (batched screening): integer significance threshold per row / column
  size_t wmaxb, cmaxu;                        // ... largest -stabiliser*log2e of every 32-row block / 32-column unit
  size_t common_total;
  // ---- dense / exact-screening / conf_matrix only ----
  size_t hi0, lo0, hi1, lo1;                  // float16 planes
  size_t f16inv;                              // [N] 1 / (power-of-two scales of the two images' float16 planes)
  size_t rowB, colB;                          // partial sum-exp of the dense sum kernel: rows [N][splits][Lp],
                                              // columns [N][panels][Sp] (one partial per workgroup)
  size_t rsum, csum;                          // softmax denominators per row / column
  size_t nmr2, nmc2;                          // nmr - log2(rsum), nmc - log2(csum): log-softmax offsets
  size_t cand_j_b, cand_x_b, ccand_i_b, ccand_x_b;   // ... the dense kernel's candidate set (samples it redid)
  size_t total;
};

// splits of the column sweep so that panels*splits*N fills the chip once
int choose_splits(int N, int panels, int tiles, int target = 256);
// (alone: FM_MODE_ALONE - launch geometry only, the offsets and sizes do not depend on it)
CoarseWs coarse_layout(int N, int L, int S, int C, int slots, bool alone = false);

struct Scalars {          // lives at ws.scalars (zeroed per call)
  unsigned flags;         // FM_DEV_* bits
  int dense_units;        // 32x32 units the sparse sum kernel left to the dense one (0: that kernel exits at once)
  int reserved;
};

// ---- launchers (each enqueues on `st`, returns hipGetLastError()) ----
hipError_t launch_prep(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                       int exact_step, int planes, hipStream_t st);
// FM_MODE_FLAT: stabilisers of every row / column, the pair margin, the float16 planes' scale; flags every sample for
// the dense sum kernel (what the screening kernel does besides screening)
hipError_t launch_stab(const CoarseWs& w, char* base, float inv_ct, float thr, int allow_dead, hipStream_t st);
hipError_t launch_prep_f16(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                           int force, hipStream_t st);
hipError_t launch_max_i8(const CoarseWs& w, char* base, hipStream_t st);
// (conf != NULL: the CONF variant - writes the dense conf_matrix of every sample from the log-softmax offsets;
// rescreen: the exact re-screening of FM_MODE_EXACT_SCREENING with the same offsets)
hipError_t launch_dense(const CoarseWs& w, char* base, float inv_ct, float thr, hipStream_t st, float* conf = nullptr,
                        int rescreen = 0);
hipError_t launch_reduce(int mode, const CoarseWs& w, char* base, float inv_ct, hipStream_t st);
// side job of the assignment launch (fm_coarse_match_maps): channels-last copy of a float32 NCHW map, or src == NULL
struct MapCopyJob {
  const float* src; float* dst; int N, Hf, Wf;
};
hipError_t launch_select(const CoarseWs& w, char* base, int h0c, int w0c, int h1c, int w1c, float inv_ct, float thr, int border,
                         float scale_px, const float* scale0, const float* scale1,
                         int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* k0, float* k1,
                         float* mconf, int cap, int32_t* d_count, int mode, hipStream_t st,
                         const MapCopyJob* job = nullptr);
hipError_t launch_conf_patch(const CoarseWs& w, char* base, float inv_ct, float* conf, hipStream_t st);
hipError_t launch_exact_lists(const CoarseWs& w, char* base, float inv_ct, const void* feat0, const void* feat1, int in_dtype,
                              int c_in, hipStream_t st);
// (the batched form of the screening - k_thresh + k_screen_rows, one wave per (row block, 64 units) - is chosen inside
// launch_screen when the batch alone fills the chip with waves)
hipError_t launch_screen(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                             float inv_ct, float thr, int dense_enabled, int allow_dead, hipStream_t st);

// Raises a kernel's dynamic-LDS limit once per (kernel, device) instead of on every launch: the
// attribute call costs tens of host microseconds, which an eager (non-graph) caller would pay per step.
template <typename K>
inline hipError_t ensure_dynamic_lds(K kernel, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
  return e;
}

}  // namespace fm
