// Host side of the C ABI (include/fmatch.h): argument checks, workspace layout, launch order.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fm_debug.h"
#include "fm_internal.h"

namespace fm {

int choose_splits(int N, int panels, int tiles, int target) {
  // One workgroup per (sample, panel, split); aim at one full round of the 256 CUs when the
  // batch alone cannot fill them (a split shorter than 2 tiles is not worth its prologue).
  const int wg = N * panels;
#ifdef FM_TUNE_ENV
  if (const char* e = getenv("FM_TARGET_WGS")) target = atoi(e) > 0 ? atoi(e) : 256;
#endif
  int s = target / (wg > 0 ? wg : 1);
  if (s < 1) s = 1;
  const int smax = tiles / 2 > 0 ? tiles / 2 : 1;
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s;
}

// Items of the screening kernel k_screen_rows: one wave per (32-row block, chunk of <= 64 column units).  64 units when the
// launch holds thousands of items anyway (a batch, a 1024x1024 pair); 32 - shorter waves, more of them - for a pair or two
// (one 640x480 pair on 4 streams: 27.0 / 28.0 / 27.3 / 26.1 k pairs/s at 64 / 32 / 16 / 8 units; 64 pairs: 286 / 309 /
// 375 us at 64 / 32 / 16)
static int choose_screen_chunks(int N, int Lp, int nunits, int* units_per_chunk, bool alone) {
  const long nrb = (long)N * (Lp / 32);
  // (FM_MODE_ALONE, a pair or two: 8-unit chunks - 3.5 us faster alone at one 640x480 pair, 7 % slower on four streams)
  // (round 6: 64-unit chunks from 1024 items on - a 640x960 pair, 1520 items: 13 624 against 13 306 pairs/s at 32 units)
  const int cu = nrb * ((nunits + 63) / 64) >= 1024 ? 64 : (alone && nrb * ((nunits + 7) / 8) <= 8192 ? 8 : 32);
  *units_per_chunk = cu;
  return (nunits + cu - 1) / cu;
}

// k_max_i8's workgroups are 4 waves (256 rows x a range of tiles) and two of them fit a CU.  When the batch alone cannot
// fill the chip the grid aims at ONE workgroup per compute unit, not two: alone the kernel is 2.8 us slower at one
// 640x480 pair (15.4 against 12.7 us), but on the bench's four streams the other half of every compute unit's registers and
// LDS is what the other pairs' kernels run in - 128 / 192 / 256 / 384 / 512 workgroups: 28.4 / 28.9 / 28.7 / 28.3 / 27.9 k
// pairs/s (round 5, with k_screen_rows; round 4's measurement with the 128-KiB-of-LDS screening kernel beside it saw no
// difference).  A batch that fills the resident slots by itself gets ~10 rounds of 512.
constexpr int kMaxPassTarget = 256;
constexpr int kMaxPassSlots = 512;

CoarseWs coarse_layout(int N, int L, int S, int C, int slots, bool alone) {
  CoarseWs w;
  memset(&w, 0, sizeof(w));
  C = padded_channels(C);
  w.N = N; w.L = L; w.S = S; w.C = C; w.slots = slots;
  w.Lp = round_up(L, kPanelRows);
  w.Sp = round_up(S, kTileCols);
  w.panels = w.Lp / kPanelRows;
  w.tiles = w.Sp / kTileCols;
  w.splits = choose_splits(N, w.panels, w.tiles);
  // (a batch that fills the resident slots by itself gets ~10 rounds of workgroups, so that the last round's tail is a
  // small share of the launch: 1216 workgroups on 512 slots were "2.4 of 3 rounds"; 64 pairs of 640x480: 351 -> 333 us
  // with 4 splits; 2 / 3 / 4 / 5 / 8 splits: 335 / 340 / 333 / 348 / 363 us)
  {
    int target = N * w.panels >= kMaxPassSlots ? 10 * kMaxPassSlots : (alone ? kMaxPassSlots : kMaxPassTarget);
    // (round 6: a launch whose workgroups would each sweep >= 32 tiles at one workgroup per CU - a 1024x1024 pair: 64 -
    // is long enough for the matrix cores to decide, and one wave per SIMD runs them at 40 %: two workgroups per CU there.
    // 1024x1024 on four streams, 256 / 384 / 512 / 768 workgroups: 6228 / 6307 / 6364 / 6207 pairs/s, max pass alone 100 /
    // 91 / 77 / 88 us; 640x960 (25 tiles per workgroup) keeps 256: 12 949 against 12 792 pairs/s)
    if (target == kMaxPassTarget && w.tiles / choose_splits(N, w.panels, w.tiles, kMaxPassTarget) >= 32) target = kMaxPassSlots;
    w.splits0 = choose_splits(N, w.panels, w.tiles, target);
  }
#ifdef FM_TUNE_ENV
  if (const char* e = getenv("FM_TARGET_WGS0")) w.splits0 = choose_splits(N, w.panels, w.tiles, atoi(e) > 0 ? atoi(e) : kMaxPassTarget);
#endif
  w.splits_s = choose_screen_chunks(N, w.Lp, w.Sp / 32, &w.units_s, alone);
  const size_t rows = (size_t)N * w.Lp, cols = (size_t)N * w.Sp;
  const size_t nblk = (rows * slots + 255) / 256;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o = align256(o + bytes); return at; };
  // ---- the common path ----
  w.zero_begin = o;
  w.cand_count = take(rows * 4);
  w.ccand_count = take(cols * 4);
  w.cand_count_b = take(rows * 4);
  w.ccand_count_b = take(cols * 4);
  w.dense_cnt = take((size_t)N * 4);
  w.cell0 = take(rows * 4);
  w.cell1 = take(cols * 4);
  w.ties0 = take((kTieCap + 1) * 4);
  w.ties1 = take((kTieCap + 1) * 4);
  w.rowmax_u = take(rows * 4);
  w.colmax_u = take(cols * 4);
  w.blocktot = take(nblk * 4);
  w.scalars = take(sizeof(Scalars));
  w.zero_end = o;
  w.q0 = take(rows * C); w.q1 = take(cols * C);
  w.sigimg = take((size_t)N * 2 * 4);
  w.imgstat = take((size_t)N * 8 * 4);
  w.l1_0 = take(rows * 4); w.l1_1 = take(cols * 4);
  w.bstat0 = take(rows / 32 * 16); w.bstat1 = take(cols / 32 * 16);
  w.emarg = take((size_t)N * 4);
  w.rowS = take(0); w.colS = take(0);
  w.nmr = take(rows * 4); w.nmc = take(cols * 4);
  w.umax = take(rows / 32 * (cols / N / 32) * 4);
  w.cand_j = take(rows * slots * 4); w.cand_x = take(rows * slots * 4);
  w.ccand_i = take(cols * slots * 4); w.ccand_x = take(cols * slots * 4);
  w.thr_r = take(rows * 4); w.thr_c = take(cols * 4);
  w.wmaxb = take(rows / 32 * 4); w.cmaxu = take(cols / 32 * 4);
  w.common_total = o;
  // ---- FM_MODE_DENSE / FM_MODE_EXACT_SCREENING / conf_matrix ----
  w.hi0 = take(rows * C * 2); w.lo0 = take(rows * C * 2);
  w.hi1 = take(cols * C * 2); w.lo1 = take(cols * C * 2);
  w.f16inv = take((size_t)N * 4);
  w.rowB = take(rows * w.splits * 4); w.colB = take(cols * w.panels * kColParts * 4);
  w.rsum = take(rows * 4); w.csum = take(cols * 4);
  w.nmr2 = take(rows * 4); w.nmc2 = take(cols * 4);
  w.cand_j_b = take(rows * slots * 4); w.cand_x_b = take(rows * slots * 4);
  w.ccand_i_b = take(cols * slots * 4); w.ccand_x_b = take(cols * slots * 4);
  w.total = o;
  return w;
}

}  // namespace fm

using namespace fm;

extern "C" int fm_version(void) { return FM_VERSION; }

extern "C" const char* fm_strerror(int s) {
  switch (s) {
    case FM_OK: return "ok";
    case FM_E_NULL: return "required pointer is NULL";
    case FM_E_SHAPE: return "inconsistent or non-positive shape";
    case FM_E_UNSUPPORTED: return "unsupported configuration (C % 4 == 0 and C <= 256, Cf = 64, W in {5,7}, thr in (0,1))";
    case FM_E_WORKSPACE: return "workspace too small or not 256-byte aligned";
    case FM_E_CAPACITY: return "more matches than the output capacity";
    case FM_E_CANDIDATES: return "a coarse row or column exceeded its candidate slots (FM_MODE_EXACT_SCREENING, then more cand_slots)";
    case FM_E_RANGE: return "descriptor not finite or |x| >= 32768, or similarities of several thousand (screening margin >= 2^60)";
    case FM_E_DENSE: return "flat similarity in a sample: call again with FM_MODE_DENSE";
    case FM_E_INTERNAL: return "assignment kernel: bounded wait for predecessor workgroups ran out; call again";
    case FM_E_STEP: return "int8 screening step too small for a descriptor outside the sampled rows: call again with FM_MODE_EXACT_STEP";
    default: return s > 0 ? hipGetErrorString((hipError_t)s) : "unknown fmatch status";
  }
}

extern "C" int fm_default_cand_slots(float thr) {
  if (!(thr > 0.f)) return 64;
  int need = (int)ceilf(1.0f / thr) + 3;
  int s = 8;
  while (s < need && s < 64) s <<= 1;
  return s;
}

// a power of two in [4, 64]: a row's slots are adjacent lanes of one wave and k_keep_emit keeps 256/slots <= 64 rows
static bool valid_slots(int s) { return s >= 4 && s <= 64 && (s & (s - 1)) == 0; }

constexpr int kKnownModes = FM_MODE_DENSE | FM_MODE_EXACT_SCREENING | FM_MODE_NO_CELL_MAPS | FM_MODE_EXACT_STEP | FM_MODE_STATS |
                            FM_MODE_FLAT | FM_MODE_ALONE;

static bool needs_dense_region(int mode, bool want_conf) {
  return want_conf || (mode & (FM_MODE_DENSE | FM_MODE_EXACT_SCREENING | FM_MODE_STATS | FM_MODE_FLAT)) != 0;
}

extern "C" int fm_coarse_workspace_bytes_mode(int N, int L, int S, int C, int cand_slots, int mode, int want_conf_matrix,
                                              size_t* bytes) {
  if (!bytes) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  if (!valid_channels(C) || !valid_slots(cand_slots)) return FM_E_UNSUPPORTED;
  if (mode & ~kKnownModes) return FM_E_UNSUPPORTED;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  *bytes = needs_dense_region(mode, want_conf_matrix != 0) ? w.total : w.common_total;
  return FM_OK;
}

extern "C" int fm_coarse_workspace_bytes(int N, int L, int S, int C, int cand_slots, size_t* bytes) {
  return fm_coarse_workspace_bytes_mode(N, L, S, C, cand_slots, FM_MODE_DENSE | FM_MODE_EXACT_SCREENING, 1, bytes);
}

static int check_coarse_shape(int N, int L, int S, int C, int cand_slots) {
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  if (!valid_channels(C) || !valid_slots(cand_slots)) return FM_E_UNSUPPORTED;
  return FM_OK;
}

// Diagnostic: the workspace layout (ints then byte offsets), so that tests can inspect the
// intermediate statistics of a run.  out[0..9] = N,L,S,C,Lp,Sp,panels,tiles,splits,slots;
// out[10..] = cand_count, ccand_count, scalars, blocktot, hi0, lo0, hi1, lo1, q0, q1, sigimg, l1_0,
// rowS, colS, rowB, colB, nmr, nmc, rsum, csum, cand_j, cand_x, ccand_i, umax, dense_cnt, rowmax_u,
// colmax_u, splits_s, units_s, total; out[40] = common_total (when n_out > 40)  (40 or 41 values).
extern "C" int fm_debug_coarse_layout(int N, int L, int S, int C, int cand_slots, int64_t* out, int n_out) {
  if (!out) return FM_E_NULL;
  if (n_out < 40) return FM_E_SHAPE;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  const int64_t v[41] = {w.N, w.L, w.S, w.C, w.Lp, w.Sp, w.panels, w.tiles, w.splits, w.slots,
                         (int64_t)w.cand_count, (int64_t)w.ccand_count, (int64_t)w.scalars, (int64_t)w.blocktot,
                         (int64_t)w.hi0, (int64_t)w.lo0, (int64_t)w.hi1, (int64_t)w.lo1, (int64_t)w.q0,
                         (int64_t)w.q1, (int64_t)w.sigimg, (int64_t)w.l1_0, (int64_t)w.rowS, (int64_t)w.colS,
                         (int64_t)w.rowB, (int64_t)w.colB, (int64_t)w.nmr, (int64_t)w.nmc, (int64_t)w.rsum,
                         (int64_t)w.csum, (int64_t)w.cand_j, (int64_t)w.cand_x, (int64_t)w.ccand_i,
                         (int64_t)w.umax, (int64_t)w.dense_cnt, (int64_t)w.rowmax_u, (int64_t)w.colmax_u,
                         w.splits_s, w.units_s, (int64_t)w.total, (int64_t)w.common_total};
  for (int i = 0; i < 40; ++i) out[i] = v[i];
  if (n_out > 40) out[40] = v[40];
  return FM_OK;
}

extern "C" int fm_coarse_match(const float* feat0, const float* feat1, int N, int L, int S, int C, int h0c, int w0c,
                               int h1c, int w1c, float temperature, float thr, int border_rm, float scale_px,
                               const float* scale0, const float* scale1, void* workspace, size_t workspace_bytes,
                               int cand_slots, int mode, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                               float* mkpts0_c, float* mkpts1_c, float* mconf, int cap, int32_t* d_count,
                               float* conf_matrix, void* stream) {
  return fm_coarse_match_dtype(feat0, feat1, FM_F32, N, L, S, C, h0c, w0c, h1c, w1c, temperature, thr, border_rm, scale_px,
                               scale0, scale1, workspace, workspace_bytes, cand_slots, mode, b_ids, i_ids,
                               j_ids, mkpts0_c, mkpts1_c, mconf, cap, d_count, conf_matrix, stream);
}

static int coarse_match_impl(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                             int h0c, int w0c, int h1c, int w1c, float temperature, float thr, int border_rm,
                             float scale_px, const float* scale0, const float* scale1, void* workspace,
                             size_t workspace_bytes, int cand_slots, int mode, int64_t* b_ids,
                             int64_t* i_ids, int64_t* j_ids, float* mkpts0_c, float* mkpts1_c, float* mconf,
                             int cap, int32_t* d_count, float* conf_matrix, const MapCopyJob* job, void* stream);

extern "C" int fm_coarse_match_dtype(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                                     int h0c, int w0c, int h1c, int w1c, float temperature, float thr, int border_rm,
                                     float scale_px, const float* scale0, const float* scale1, void* workspace,
                                     size_t workspace_bytes, int cand_slots, int mode, int64_t* b_ids,
                                     int64_t* i_ids, int64_t* j_ids, float* mkpts0_c, float* mkpts1_c, float* mconf,
                                     int cap, int32_t* d_count, float* conf_matrix, void* stream) {
  return coarse_match_impl(feat0, feat1, in_dtype, N, L, S, C, h0c, w0c, h1c, w1c, temperature, thr, border_rm, scale_px,
                           scale0, scale1, workspace, workspace_bytes, cand_slots, mode, b_ids, i_ids, j_ids, mkpts0_c,
                           mkpts1_c, mconf, cap, d_count, conf_matrix, nullptr, stream);
}

// fm_coarse_match_dtype + the channels-last copy of image 1's fine map as a side job of the assignment launch
extern "C" int fm_coarse_match_maps(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                                    int h0c, int w0c, int h1c, int w1c, float temperature, float thr, int border_rm,
                                    float scale_px, const float* scale0, const float* scale1, void* workspace,
                                    size_t workspace_bytes, int cand_slots, int mode, int64_t* b_ids,
                                    int64_t* i_ids, int64_t* j_ids, float* mkpts0_c, float* mkpts1_c, float* mconf,
                                    int cap, int32_t* d_count, float* conf_matrix, const float* feat_f1, int Nf, int Cf,
                                    int Hf1, int Wf1, void* scratch1, void* stream) {
  if (!feat_f1 || !scratch1) return FM_E_NULL;
  if (Nf <= 0 || Hf1 <= 0 || Wf1 <= 0) return FM_E_SHAPE;
  if (Cf != 64) return FM_E_UNSUPPORTED;
  if (((uintptr_t)scratch1 & 15) || ((uintptr_t)feat_f1 & 15)) return FM_E_WORKSPACE;
  const MapCopyJob job{feat_f1, (float*)scratch1, Nf, Hf1, Wf1};
  return coarse_match_impl(feat0, feat1, in_dtype, N, L, S, C, h0c, w0c, h1c, w1c, temperature, thr, border_rm, scale_px,
                           scale0, scale1, workspace, workspace_bytes, cand_slots, mode, b_ids, i_ids, j_ids, mkpts0_c,
                           mkpts1_c, mconf, cap, d_count, conf_matrix, &job, stream);
}

static int coarse_match_impl(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                             int h0c, int w0c, int h1c, int w1c, float temperature, float thr, int border_rm,
                             float scale_px, const float* scale0, const float* scale1, void* workspace,
                             size_t workspace_bytes, int cand_slots, int mode, int64_t* b_ids,
                             int64_t* i_ids, int64_t* j_ids, float* mkpts0_c, float* mkpts1_c, float* mconf,
                             int cap, int32_t* d_count, float* conf_matrix, const MapCopyJob* job, void* stream) {
  if (!feat0 || !feat1 || !workspace || !d_count) return FM_E_NULL;
  if (in_dtype != FM_F32 && in_dtype != FM_F16 && in_dtype != FM_BF16) return FM_E_UNSUPPORTED;
  if (cap > 0 && (!b_ids || !i_ids || !j_ids || !mkpts0_c || !mkpts1_c || !mconf)) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0 || cap < 0 || L != h0c * w0c || S != h1c * w1c) return FM_E_SHAPE;
  if (!valid_channels(C) || !valid_slots(cand_slots)) return FM_E_UNSUPPORTED;
  if (!(thr > 0.f) || !(thr < 1.f) || !(temperature > 0.f)) return FM_E_UNSUPPORTED;
  if (mode & ~kKnownModes) return FM_E_UNSUPPORTED;
  const bool exact = (mode & FM_MODE_EXACT_SCREENING) != 0;
  const bool dense = needs_dense_region(mode, conf_matrix != nullptr);      // exact screening and conf_matrix read the planes too
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots, (mode & FM_MODE_ALONE) != 0);
  if (workspace_bytes < (dense ? w.total : w.common_total) || ((uintptr_t)workspace & 255)) return FM_E_WORKSPACE;
  char* base = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const float inv_ct = 1.0f / ((float)C * temperature);
  const float thr_list = conf_matrix ? fminf(thr, 0.1f) : thr;      // candidate threshold of the dense kernels' LISTS

  // The common path is four launches: prep -> max pass -> sparse sum kernel -> assignment.
  // one dispatch: clear the per-call counters, quantise both images (one int8 step per image), L1 norms
  // (FM_MODE_EXACT_STEP: the images' largest |x| first - a memset node and one small kernel - and the int8 step from them)
  // (FM_MODE_FLAT: the float16 planes of every sample too - the dense sum kernel will want them all)
  // (a hint: not taken when the dense conf_matrix / the softmax statistics are wanted - their exact rewrite reads the
  // screening kernel's lists)
  const bool flat = (mode & FM_MODE_FLAT) != 0 && !conf_matrix && !(mode & FM_MODE_STATS);
  hipError_t e = launch_prep(feat0, feat1, in_dtype, C, w, base, (mode & FM_MODE_EXACT_STEP) ? 1 : 0, flat ? 1 : 0, st);
  if (e != hipSuccess) return (int)e;
  // max pass: row / column / unit maxima of the integer screening product (atomicMax: no partials, no reduction kernel)
  e = launch_max_i8(w, base, st);
  if (e != hipSuccess) return (int)e;
  // sparse sum kernel: stabilisers, live units, exact terms of the few significant entries, candidates (listed per row
  // and per column); flags the samples with too many significant entries per unit (flat similarity)
  // (dead-row certificates only when nobody reads every row's denominator: the dense conf_matrix does)
  const bool stats = (mode & FM_MODE_STATS) != 0;      // the softmax statistics of EVERY row and column are wanted
  // (FM_MODE_FLAT: the caller expects flat similarity everywhere - the sweep would only find that out again; a small
  // kernel forms the stabilisers and flags every sample for the dense sum kernel)
  if (flat) e = launch_stab(w, base, inv_ct, thr, (conf_matrix || stats) ? 0 : 1, st);
  else e = launch_screen(feat0, feat1, in_dtype, C, w, base, inv_ct, thr, dense ? 1 : 0, (conf_matrix || stats) ? 0 : 1, st);
  if (e != hipSuccess) return (int)e;
  if (dense && !flat) {
    // float16 hi / lo planes for the samples that go on to the dense kernel (all of them when the exact screening or
    // the conf_matrix sweep will run); exits at once otherwise
    e = launch_prep_f16(feat0, feat1, in_dtype, C, w, base, exact ? 1 : (conf_matrix ? 2 : 0), st);
    if (e != hipSuccess) return (int)e;
  }
  if (dense) {
    // dense sum kernel (float32-equivalent hi/lo product on the matrix cores): redoes the samples in which the sparse
    // kernel flagged units (one arithmetic per sample keeps exact conf ties exact); exits at once when there are none
    // (a conf_matrix request: candidates down to min(thr, 0.1) - the lists then name every entry of a dense sample with
    // conf > 0.1, which k_exact_lists resolves from exact dot products; the assignment applies thr itself)
    e = launch_dense(w, base, inv_ct, thr_list, st);
    if (e != hipSuccess) return (int)e;
  }
  // The assignment folds the softmax denominators of its candidates from the partial sums itself.  The
  // denominators / log-softmax offsets of EVERY row and column are only needed by the exact screening and by the
  // dense conf_matrix:
  // (FM_MODE_FLAT with more than the default 8 candidate slots - rows without a peak next to peaked ones: every sample's
  // denominators are folds of the dense kernel's 13 + 19 partials, and the assignment would redo them for every
  // candidate and every competing row - 35 us at 16 slots against 21 us with one reduction launch (5 us) in front; at 8
  // slots the rows hold one or two candidates and the launch costs more than it saves)
  const bool reduced = exact || conf_matrix || stats || (flat && cand_slots > 8);   // (the assignment then reads the folded denominators)
  if (reduced) {
    e = launch_reduce(1, w, base, inv_ct, st);
    if (e != hipSuccess) return (int)e;
  }
  if (exact) {                 // exits immediately unless the sum kernels' screening overflowed a row's slots
    e = launch_dense(w, base, inv_ct, thr_list, st, nullptr, 1);
    if (e != hipSuccess) return (int)e;
  }
  if ((conf_matrix || stats) && dense) {  // every entry is read: the dense kernel's lists and their denominators made exact together
    e = launch_exact_lists(w, base, inv_ct, feat0, feat1, in_dtype, C, st);
    if (e != hipSuccess) return (int)e;
  }
  if (conf_matrix) {           // dense data['conf_matrix'] on request (one more sweep)
    e = launch_dense(w, base, inv_ct, thr, st, conf_matrix);
    if (e != hipSuccess) return (int)e;
    // ... whose hi/lo-split products carry 22 bits: the entries that matter are rewritten from their exact float32 dot
    // products (the rows' lists of significant entries)
    e = launch_conf_patch(w, base, inv_ct, conf_matrix, st);
    if (e != hipSuccess) return (int)e;
  }
  e = launch_select(w, base, h0c, w0c, h1c, w1c, inv_ct, thr, border_rm, scale_px, scale0, scale1,
                    b_ids, i_ids, j_ids, mkpts0_c, mkpts1_c, mconf, cap, d_count,
                    (dense ? (mode | FM_MODE_DENSE) : mode) | (reduced ? FM_MODE_STATS : 0), st, job);
  return (int)e;
}

// ---------------------------------------------------------------------------------------------------------------------
// fm_coarse_match_auto: ONE call for any data (the reference's CoarseMatching.forward is one call,
// network/utils/coarse_matching_new.py:43-73).  The data-dependent conditions the device reports - flat similarity,
// candidate-slot overflow, a clipped int8 step, the assignment's bounded wait - are answered here, on the host, behind
// the host sync the reference has too (torch.where, :109); what is left for the caller is FM_E_CAPACITY (its output
// buffers are too small: *m_out = the capacity needed), FM_E_RANGE (bad input) and argument errors.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kAutoDataModes = FM_MODE_DENSE | FM_MODE_EXACT_SCREENING | FM_MODE_EXACT_STEP | FM_MODE_FLAT;

// FM_E_DENSE after a call on the common path: the max pass, the screening kernel's lists and its per-sample flags in
// the workspace are all valid - what is missing is the dense sum kernel's work on the flagged samples and an
// assignment that reads it.  Continue from there instead of repeating prep, max pass and screening.
int resume_with_dense(const void* feat0, const void* feat1, int in_dtype, int C, const CoarseWs& w, char* base,
                      int h0c, int w0c, int h1c, int w1c, float inv_ct, float thr, int border_rm, float scale_px,
                      const float* scale0, const float* scale1, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                      float* mkpts0_c, float* mkpts1_c, float* mconf, int cap, int32_t* d_count, int mode, hipStream_t st) {
  // what the first assignment launch left behind: its cell maps and tie lists, the workgroup totals of its look-back;
  // and the FM_DEV_DENSE bit (dense_units, next to it, stays: the dense kernels read it)
  hipError_t e = hipMemsetAsync(base + w.cell0, 0, w.rowmax_u - w.cell0, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(base + w.blocktot, 0, w.scalars - w.blocktot, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(base + w.scalars, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  e = launch_prep_f16(feat0, feat1, in_dtype, C, w, base, 0, st);
  if (e != hipSuccess) return (int)e;
  e = launch_dense(w, base, inv_ct, thr, st);
  if (e != hipSuccess) return (int)e;
  return (int)launch_select(w, base, h0c, w0c, h1c, w1c, inv_ct, thr, border_rm, scale_px, scale0, scale1, b_ids, i_ids,
                            j_ids, mkpts0_c, mkpts1_c, mconf, cap, d_count, mode | FM_MODE_DENSE, st, nullptr);
}
}  // namespace

extern "C" int fm_coarse_workspace_bytes_auto(int N, int L, int S, int C, int max_cand_slots, size_t* bytes) {
  if (max_cand_slots == 0) max_cand_slots = 64;
  return fm_coarse_workspace_bytes(N, L, S, C, max_cand_slots, bytes);
}

extern "C" int fm_coarse_match_auto(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                                    int h0c, int w0c, int h1c, int w1c, float temperature, float thr, int border_rm,
                                    float scale_px, const float* scale0, const float* scale1, void* workspace,
                                    size_t workspace_bytes, int max_cand_slots, int mode, int64_t* b_ids, int64_t* i_ids,
                                    int64_t* j_ids, float* mkpts0_c, float* mkpts1_c, float* mconf, int cap,
                                    int32_t* d_count, float* conf_matrix, int32_t* hint_io, int32_t* m_out,
                                    int32_t* info_out, void* stream) {
  if (!m_out) return FM_E_NULL;
  if (max_cand_slots == 0) max_cand_slots = 64;
  if (!valid_slots(max_cand_slots)) return FM_E_UNSUPPORTED;
  if (mode & ~kKnownModes) return FM_E_UNSUPPORTED;
  if (!(thr > 0.f) || !(thr < 1.f)) return FM_E_UNSUPPORTED;
  const int slots0 = fm_default_cand_slots(thr) < max_cand_slots ? fm_default_cand_slots(thr) : max_cand_slots;
  // the caller's fixed options (cell maps, statistics) stay; the data-dependent bits it passes are a starting point
  const int fixed = mode & ~kAutoDataModes;
  const bool full_stats = conf_matrix != nullptr || (mode & FM_MODE_STATS) != 0;
  int cur = mode & kAutoDataModes;
  if (full_stats) cur |= FM_MODE_EXACT_SCREENING;        // (that path runs the denominator reduction the re-screening needs anyway)
  int slots = slots0;
  if (conf_matrix && slots < 16 && max_cand_slots >= 16) slots = 16;     // (its dense lists go down to conf 0.1: <= 10 + band per row)
  const int slots_start = slots;                         // what this REQUEST starts with: not something the data taught
  if (hint_io && *hint_io) {                             // what served the previous call of this kind
    cur |= *hint_io & kAutoDataModes;
    const int hs = (*hint_io >> 8) & 0xff;
    if (valid_slots(hs) && hs <= max_cand_slots && hs > slots) slots = hs;
  }
  bool tried_wide = slots > slots0, retried_internal = false;
  int st = FM_OK;
  int32_t m = 0, info = 0;
  int attempts = 0;
  for (int attempt = 0; attempt < 10; ++attempt) {
    ++attempts;
    const int call_mode = fixed | cur | ((cur & FM_MODE_FLAT) ? FM_MODE_DENSE : 0);
    // every attempt must fit the caller's workspace (sized by fm_coarse_workspace_bytes_auto for max_cand_slots)
    st = fm_coarse_match_dtype(feat0, feat1, in_dtype, N, L, S, C, h0c, w0c, h1c, w1c, temperature, thr, border_rm, scale_px,
                               scale0, scale1, workspace, workspace_bytes, slots, call_mode, b_ids, i_ids, j_ids, mkpts0_c,
                               mkpts1_c, mconf, cap, d_count, conf_matrix, stream);
    if (st != FM_OK) return st;
    st = fm_read_count_info(d_count, cap, &m, &info, stream);
    if (st == FM_E_DENSE && !(cur & FM_MODE_DENSE) && !conf_matrix && !(mode & FM_MODE_STATS)) {
      // the common path's own results are still in the workspace: add the dense kernels' part and assign again
      cur |= FM_MODE_DENSE;
      const CoarseWs w = coarse_layout(N, L, S, C, slots);
      // (the resume clears the WHOLE status word: only when "flat similarity" is all it holds - anything else the device
      // reported with it would be dropped, so such a call is repeated from the start instead)
      if (workspace_bytes >= w.total && (info & ~FM_DEV_ALL_DENSE) == FM_DEV_DENSE) {
        st = resume_with_dense(feat0, feat1, in_dtype, C, w, (char*)workspace, h0c, w0c, h1c, w1c,
                               1.0f / ((float)C * temperature), thr, border_rm, scale_px, scale0, scale1, b_ids, i_ids, j_ids,
                               mkpts0_c, mkpts1_c, mconf, cap, d_count, fixed | cur, (hipStream_t)stream);
        if (st != FM_OK) return st;
        st = fm_read_count_info(d_count, cap, &m, &info, stream);
      } else {
        continue;
      }
    }
    if (st == FM_OK) break;
    if (st == FM_E_STEP && !(cur & FM_MODE_EXACT_STEP)) { cur |= FM_MODE_EXACT_STEP; continue; }
    if (st == FM_E_DENSE && !(cur & FM_MODE_DENSE)) { cur |= FM_MODE_DENSE; continue; }
    if (st == FM_E_CANDIDATES && (cur & FM_MODE_DENSE) && !tried_wide && slots < 16 && max_cand_slots >= 16 &&
        !(cur & FM_MODE_EXACT_SCREENING)) {
      // rows without a peak next to peaked ones hold more near-candidates than the default slots: twice the slots and
      // the int8 step from the images' true maxima (margins 1.5x narrower) before the exact re-screening sweep
      slots = 16; cur |= FM_MODE_EXACT_STEP; tried_wide = true;
      continue;
    }
    if (st == FM_E_CANDIDATES && !(cur & FM_MODE_EXACT_SCREENING)) {
      if (tried_wide) slots = slots0;                    // the wider lists did not hold them either
      cur |= FM_MODE_EXACT_SCREENING | FM_MODE_DENSE;
      continue;
    }
    if (st == FM_E_CANDIDATES && slots < max_cand_slots) { slots *= 2; continue; }
    if (st == FM_E_INTERNAL && !retried_internal) { retried_internal = true; continue; }
    break;                                               // FM_E_CAPACITY, FM_E_RANGE, what persists: the caller's
  }
  *m_out = m;
  if (info_out) *info_out = info;
  if (hint_io && (st == FM_OK || st == FM_E_CAPACITY)) {
    int learnt = cur & kAutoDataModes;
    if (full_stats) learnt &= ~FM_MODE_EXACT_SCREENING | (mode & FM_MODE_EXACT_SCREENING);   // (implied by the request, not learnt)
    // every sample went to the dense sum kernel: the next call of this kind skips the screening sweep (FM_MODE_FLAT)
    if ((learnt & FM_MODE_DENSE) && !full_stats) {
      if (info & FM_DEV_ALL_DENSE) learnt |= FM_MODE_FLAT; else learnt &= ~FM_MODE_FLAT;
    }
    // second byte: the slots the DATA asked for beyond this request's own starting point (0 = none: a conf_matrix
    // call's 16 are implied by the request and must not follow plain calls of the shape); third byte: the slot count
    // the serving attempt ran with - ALWAYS written (the workspace layout fm_coarse_cell_maps / fm_coarse_softmax_stats
    // must be asked for), ignored on input
    *hint_io = learnt | (slots != slots_start ? slots << 8 : 0) | (slots << 16) | (attempts << 24);
  }
  return st;
}

// Device pointers of the cell -> (match index + 1) maps the coarse stage leaves in its workspace
// (0 = cell unmatched; pitch = padded cells per sample).  fm_gather_windows_cells consumes them.
extern "C" int fm_coarse_cell_maps(void* workspace, int N, int L, int S, int C, int cand_slots, int32_t** cell0,
                                   int* pitch0, int32_t** ties0, int32_t** cell1, int* pitch1, int32_t** ties1) {
  if (!workspace || !cell0 || !cell1 || !pitch0 || !pitch1 || !ties0 || !ties1) return FM_E_NULL;
  if (N <= 0 || L <= 0 || S <= 0) return FM_E_SHAPE;
  if (!valid_channels(C) || !valid_slots(cand_slots)) return FM_E_UNSUPPORTED;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  *cell0 = (int32_t*)((char*)workspace + w.cell0); *pitch0 = w.Lp; *ties0 = (int32_t*)((char*)workspace + w.ties0);
  *cell1 = (int32_t*)((char*)workspace + w.cell1); *pitch1 = w.Sp; *ties1 = (int32_t*)((char*)workspace + w.ties1);
  return FM_OK;
}

// Device pointers of the softmax statistics the coarse stage leaves in its workspace when it ran with FM_MODE_STATS or a
// conf_matrix request: softmax(sim, dim 2)[b,i,j] = exp2(k2 x + nm_r[b * pitch_r + i]) / sum_r[b * pitch_r + i] and
// softmax(sim, dim 1)[b,i,j] = exp2(k2 x + nm_c[b * pitch_c + j]) / sum_c[b * pitch_c + j] with x = feat0[b,i] . feat1[b,j]
// and k2 = log2(e) / (C temperature).
extern "C" int fm_coarse_softmax_stats(void* workspace, int N, int L, int S, int C, int cand_slots, const float** nm_r,
                                       const float** sum_r, int* pitch_r, const float** nm_c, const float** sum_c,
                                       int* pitch_c) {
  if (!workspace || !nm_r || !sum_r || !pitch_r || !nm_c || !sum_c || !pitch_c) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  *nm_r = (const float*)((char*)workspace + w.nmr); *sum_r = (const float*)((char*)workspace + w.rsum); *pitch_r = w.Lp;
  *nm_c = (const float*)((char*)workspace + w.nmc); *sum_c = (const float*)((char*)workspace + w.csum); *pitch_c = w.Sp;
  return FM_OK;
}

// Diagnostic: launch ONE correlation sweep (mode 0 = pass A, 1 = pass B) on a workspace that a
// previous fm_coarse_match call with the same shapes has filled, so that a benchmark can bracket
// exactly that kernel with events.  Pass B's candidate counters are zeroed first (outside any
// bracket the caller places after this function's memset is enqueued... the memset precedes the kernel).
extern "C" int fm_debug_launch_corr(void* workspace, int N, int L, int S, int C, int cand_slots, float temperature,
                                    float thr, int mode, void* stream) {
  if (!workspace) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  if (mode < 0 || mode > 2) return FM_E_UNSUPPORTED;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) return (int)launch_max_i8(w, (char*)workspace, st);
  if (mode == 1) return (int)launch_dense(w, (char*)workspace, 1.0f / ((float)C * temperature), thr, st);
  return (int)launch_dense(w, (char*)workspace, 1.0f / ((float)C * temperature), thr, st, nullptr, 1);      // mode 2: the re-screening
}

// Diagnostic: launch the sparse sum kernel alone on a workspace a previous fm_coarse_match filled.
extern "C" int fm_debug_launch_screen(void* workspace, const float* feat0, const float* feat1, int N, int L, int S,
                                          int C, int cand_slots, float temperature, float thr, void* stream) {
  if (!workspace || !feat0 || !feat1) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  return (int)launch_screen(feat0, feat1, FM_F32, C, w, (char*)workspace, 1.0f / ((float)C * temperature), thr, 1, 1,
                                (hipStream_t)stream);
}

// Diagnostic: launch k_prep_split alone (it clears the per-call counters: run a complete fm_coarse_match afterwards
// before anything reads the workspace's candidate lists again).
extern "C" int fm_debug_launch_prep(void* workspace, const float* feat0, const float* feat1, int N, int L, int S, int C,
                                    int cand_slots, void* stream) {
  if (!workspace || !feat0 || !feat1) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  return (int)launch_prep(feat0, feat1, FM_F32, C, w, (char*)workspace, 0, 0, (hipStream_t)stream);
}

// Diagnostic: launch the float16 plane kernel alone (force = 1: every sample; 0: the samples flagged for the dense
// kernel) on a workspace a previous fm_coarse_match filled.
extern "C" int fm_debug_launch_prep_f16(void* workspace, const float* feat0, const float* feat1, int N, int L, int S,
                                        int C, int cand_slots, int force, void* stream) {
  if (!workspace || !feat0 || !feat1) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  return (int)launch_prep_f16(feat0, feat1, FM_F32, C, w, (char*)workspace, force, (hipStream_t)stream);
}

// Diagnostic: the two launches FM_MODE_FLAT has of its own (which = 0: k_prep_split writing the float16 planes too,
// 1: k_stab) on a workspace a previous FM_MODE_FLAT call filled.
extern "C" int fm_debug_launch_flat(void* workspace, const float* feat0, const float* feat1, int N, int L, int S, int C,
                                    int cand_slots, float temperature, float thr, int which, void* stream) {
  if (!workspace || !feat0 || !feat1) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  if (which < 0 || which > 1) return FM_E_UNSUPPORTED;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  if (which == 0) return (int)launch_prep(feat0, feat1, FM_F32, C, w, (char*)workspace, 0, 1, (hipStream_t)stream);
  return (int)launch_stab(w, (char*)workspace, 1.0f / ((float)C * temperature), thr, 1, (hipStream_t)stream);
}

// Diagnostic: zero the candidate counters and the scalars, so that the sum kernels can be launched again on a
// workspace whose max-pass results are kept.
extern "C" int fm_debug_reset_counters(void* workspace, int N, int L, int S, int C, int cand_slots, void* stream) {
  if (!workspace) return FM_E_NULL;
  const int bad = check_coarse_shape(N, L, S, C, cand_slots);
  if (bad) return bad;
  const CoarseWs w = coarse_layout(N, L, S, C, cand_slots);
  hipError_t e = hipMemsetAsync((char*)workspace + w.cand_count, 0, w.cell0 - w.cand_count, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemsetAsync((char*)workspace + w.scalars, 0, sizeof(Scalars), (hipStream_t)stream);
}

extern "C" int fm_read_count(const int32_t* d_count, int cap, int32_t* m_out, void* stream) {
  return fm_read_count_info(d_count, cap, m_out, nullptr, stream);
}

extern "C" int fm_read_count_info(const int32_t* d_count, int cap, int32_t* m_out, int32_t* info_out, void* stream) {
  if (!d_count || !m_out) return FM_E_NULL;
  int32_t h[2] = {0, 0};
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(h, d_count, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return (int)e;
  *m_out = h[0];
  if (info_out) *info_out = h[1];
  if (h[1] & FM_DEV_INTERNAL) return FM_E_INTERNAL;
  if (h[1] & FM_DEV_RANGE) return FM_E_RANGE;
  if (h[1] & FM_DEV_STEP) return FM_E_STEP;
  if (h[1] & FM_DEV_DENSE) return FM_E_DENSE;
  if (h[1] & FM_DEV_CANDIDATES) return FM_E_CANDIDATES;
  if (h[1] & FM_DEV_CAPACITY) return FM_E_CAPACITY;
  if (h[0] > cap) return FM_E_CAPACITY;
  return FM_OK;
}
