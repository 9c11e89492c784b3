// Coarse stage, sparse assignment on the candidate lists of the sum kernels - ONE kernel:
//   float32 conf of every candidate from its dot product and the softmax denominators (folded from the partial sums
//   on the spot), best conf of its row (the row's candidates sit in adjacent lanes) and of its column (the column's
//   candidates, listed per column by the sum kernels, are re-evaluated with the same arithmetic - no grid-wide pass),
//   threshold + mutual nearest neighbour + border, a row's matches sorted by j, deterministic prefix offsets
//   (look-back over the workgroup totals, bounded wait) -> outputs in (b, i, j) order.
//
// Follows network/utils/coarse_matching_new.py:99-141.  Every entry of the L x S matrix
// that is not a candidate has conf <= thr, so it can neither pass :99 nor beat a surviving
// candidate in the row / column maxima of :105-106 - the maxima over candidates decide.
// The border mask (:100-102) is applied last: border cells still compete in the maxima.
#include "fm_internal.h"
#include "fm_maps_device.h"

namespace fm {

struct SelArgs {
  const float* nmr; const float* nmc;
  const float* rowB; const float* colB;   // partial sums of the dense sum kernel (samples it redid): rows [N][splits][Lp], columns [N][panels][Sp]
  int exact;                              // exact screening ran: its overflow is then FM_DEV_CANDIDATES already
  int cell_maps;                          // write the cell -> match maps (two returning atomics per match)
  int dense_enabled;                      // the call runs the dense sum kernel (FM_MODE_DENSE): its regions exist
  const int* cand_count; const int* cand_j; const float* cand_x;          // the screening kernel's SIGNIFICANT entries, per row
  const int* ccand_count; const int* ccand_i; const float* ccand_x;       // ... and per column (k_screen: (index, exact x) lists)
  const int* cand_count_b; const int* cand_j_b; const float* cand_x_b;    // the dense one's (samples with dense_cnt > 0)
  const int* ccand_count_b; const int* ccand_i_b; const float* ccand_x_b;
  const int* dense_cnt;
  const float* rsum; const float* csum;   // softmax denominators of every row / column (k_reduce_sums), when sums_ready
  int sums_ready;                         // k_reduce_sums ran before this kernel (exact screening, conf_matrix, statistics,
                                          // FM_MODE_FLAT): a dense sample's denominators are ONE load each instead of a
                                          // fold of 13 + 19 partials per candidate and per competing row
  int* blocktot; Scalars* scal;
  int N, L, S, C, Lp, Sp, splits, splits_s, panels, slots;
  int h0c, w0c, h1c, w1c, border;
  float k, thr, scale_px;
  const float* scale0; const float* scale1;
  int64_t* b_ids; int64_t* i_ids; int64_t* j_ids; float* k0; float* k1; float* mconf;
  int cap; int32_t* d_count;
  int* cell0; int* cell1;     // cell -> match index + 1 (for the cell-ordered window gathers)
  int* ties0; int* ties1;     // [0] = count, then the matches that lost their cell to an exactly tied match
  int nblk;                   // logical blocks of 256 (row, slot) pairs
  MapCopyJob job;             // side job (fm_coarse_match_maps): workgroups nblk .. gridDim.x - 1 transpose a fine map
  float* diag;                // diagnostic build: stamp buffer
};

__device__ __forceinline__ bool interior(int id, int hh, int ww, int bd) {
  if (bd <= 0) return true;
  const int y = id / ww, x = id - y * ww;
  return y >= bd && y < hh - bd && x >= bd && x < ww - bd;
}

// n partial sums `pitch` floats apart, added in index order (the order is part of the result: every thread that
// needs a denominator folds it the same way, so equal inputs give equal bits).  Up to 32 loads in flight - the 19
// column partials of a 640x480 pair are ONE memory round trip, not three - with clamped indices instead of predicates
// (a repeated load of the last partial is discarded by the select).
__device__ __forceinline__ float fold_partials(const float* p, int n, long pitch) {
  float t = 0.f;
  for (int q0 = 0; q0 < n; q0 += 32) {
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = p[(long)min(q0 + q, n - 1) * pitch];
#pragma unroll
    for (int q = 0; q < 32; ++q) t += (q0 + q < n) ? v[q] : 0.f;
  }
  return t;
}

// Sum of the terms e of a list's live entries in the order of their keys (the lists are appended to in arrival order:
// sorting by key - the entry's column or row index, distinct inside a list - makes the float32 sum a function of the data
// only, and equal lists give equal bits).  A list sits in the `slots` adjacent lanes of its row group; rank = number of
// live entries with a smaller key; the r-th term is picked up by a group OR (every other lane contributes +0).  `maxcnt`
// = the longest list of the wave (uniform): one round for the usual one-entry lists.
__device__ __forceinline__ float list_sum(float e, int key, bool live, int slots, int base, int maxcnt) {
  if (maxcnt <= 1) return __shfl(live ? e : 0.f, base);      // slot 0 holds the one entry (or the list is empty)
  int rank = 0;
  for (int q = 0; q < slots; ++q) {
    const int ok = __shfl(key, base + q);
    const int ol = __shfl(live ? 1 : 0, base + q);
    rank += (ol && ok < key) ? 1 : 0;
  }
  float t = 0.f;
  for (int rr = 0; rr < maxcnt; ++rr) {
    unsigned v = (live && rank == rr) ? __float_as_uint(e) : 0u;
    for (int m = 1; m < slots; m <<= 1) v |= (unsigned)__shfl_xor((int)v, m);
    t += __uint_as_float(v);
  }
  return t;
}

// conf of an entry from the dot product the sum kernel produced for it - the same number that entered the row and
// the column sum, so numerator and denominator are consistent (as in the reference's softmax, :68)
__device__ __forceinline__ float entry_conf(float x, float k, float nmr, float rs, float nmc, float cs) {
#pragma clang fp contract(off)
  const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(x, k, nmr)) / rs;
  const float pc = __builtin_amdgcn_exp2f(__builtin_fmaf(x, k, nmc)) / cs;
  return pr * pc;
}

// One logical block = 256 (row, slot) pairs = 256/slots consecutive rows; a row's `slots` threads are adjacent lanes of
// one wave.  Output offsets come from a look-back over the totals of the blocks before it.  Block = blockIdx: a
// workgroup then waits for workgroups of lower index, which the dispatcher starts first in practice (in-kernel stamps:
// all 152 workgroups of a 640x480 pair start within 0.3 us); HIP does not promise that order, so the wait is BOUNDED -
// a workgroup that does not see its predecessors within ~2^22 polls gives up and the call reports FM_E_INTERNAL
// instead of hanging.  (Tickets - a returning atomic on one address before anything else can be loaded - made the
// order a certainty at 1.2 us per launch and 146 us for the 9728 blocks of a 64-pair batch.)
__global__ __launch_bounds__(256) void k_select(SelArgs a) {
  __shared__ int sm[4];
  __shared__ int rowoff[64], rowcnt[64];
  __shared__ int s_kj[256];
  __shared__ float s_kc[256];
#ifdef FM_DIAG_CLOCK       // diagnostic build only: constant-clock stamps (10 ns) of every workgroup's phases
  unsigned long long dg[6];
  dg[0] = __builtin_amdgcn_s_memrealtime();
#define SEL_STAMP(i) dg[i] = __builtin_amdgcn_s_memrealtime();
#else
#define SEL_STAMP(i)
#endif
  // ---- side-job role (fm_coarse_match_maps): the workgroups BEHIND the assignment's own - those are dispatched first,
  // their look-back chain is not delayed - copy image 1's NCHW fine map to channels-last storage: an HBM-bound stream
  // next to 152 latency-bound workgroups that leave the memory system idle ----
  if ((int)blockIdx.x >= a.nblk) {
    __shared__ float tr_tile[64 * 65];
    nchw_to_nhwc64_units<float>(a.job.src, a.job.dst, a.job.Hf, a.job.Wf, a.job.N, tr_tile, (long)blockIdx.x - a.nblk,
                                (long)gridDim.x - a.nblk);
    return;
  }
  const int blk = blockIdx.x;
  SEL_STAMP(1)
  const int lane = threadIdx.x & 63;
  const long gid = (long)blk * 256 + threadIdx.x;
  const long grow = gid / a.slots;                 // b*Lp + i
  const int slot = (int)(gid - grow * a.slots);
  const int b = (int)(grow / a.Lp);
  const int i = (int)(grow - (long)b * a.Lp);
  // a sample is handled by ONE sum kernel: the dense one redid it if the sparse one flagged any of its units.  (Without
  // FM_MODE_DENSE a flagged sample has no valid result - the call reports FM_E_DENSE - and the dense kernel's regions
  // of the workspace do not exist: they must not be touched.)
  const bool dense = a.dense_enabled && b < a.N && a.dense_cnt[b] > 0;       // (uniform: a block never straddles samples)
  const int* cand_count = dense ? a.cand_count_b : a.cand_count;
  const int* cand_j = dense ? a.cand_j_b : a.cand_j;
  const float* cand_x = dense ? a.cand_x_b : a.cand_x;
  const int* ccand_count = dense ? a.ccand_count_b : a.ccand_count;
  const int* ccand_i = dense ? a.ccand_i_b : a.ccand_i;
  const float* ccand_x = dense ? a.ccand_x_b : a.ccand_x;
  // The `slots` lanes of a row work TOGETHER on the row's entries (usually one).
  //  * dense samples: the lists hold CANDIDATES and the denominators come as partial sums of the dense kernel - lane s
  //    of the group loads the partials s, s + slots, ... and the group adds the lanes' shares in lane order (one load
  //    per lane and round trip, a fixed order that every group evaluating the same row or column reproduces);
  //  * all other samples: the lists hold every SIGNIFICANT entry of the row / column with its exact dot product
  //    (k_screen), and a denominator is the sum of its list's terms in index order (list_sum).
  const bool row_ok = b < a.N && i < a.L;
  const long grow_c = row_ok ? grow : 0;                      // (clamped: no load behind a branch)
  const int b_c = row_ok ? b : 0, i_c = row_ok ? i : 0;
  const int base = lane - slot;                               // first lane of this row's group
  auto group_sum = [&](float v) {                             // v of lane base, + base+1, ... in that order
    float t = 0.f;
    for (int q = 0; q < a.slots; ++q) t += __shfl(v, base + q);
    return t;
  };
  auto share = [&](const float* p, int n, long pitch) {       // this lane's share of n partials `pitch` apart
    float t = 0.f;
    for (int q = slot; q < n; q += a.slots) t += p[(long)q * pitch];
    return t;
  };
  auto wave_max_int = [&](int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
  };
  // the denominator of row `row` (of sample b_c): its list (lane s = entry s) or its partial sums
  auto row_denominator = [&](int row, float nm) {
    const long g = (long)b_c * a.Lp + row;
    if (dense) return a.sums_ready ? a.rsum[g] : group_sum(share(a.rowB + (long)b_c * a.splits * a.Lp + row, a.splits, a.Lp));
    const int c2 = min(cand_count[g], a.slots);
    const int k2 = cand_j[g * a.slots + slot];
    const float x2 = cand_x[g * a.slots + slot];
    return list_sum(__builtin_amdgcn_exp2f(__builtin_fmaf(x2, a.k, nm)), k2, slot < c2, a.slots, base, wave_max_int(c2));
  };
  // round trip 1: entry count, this lane's entry (speculatively), stabiliser, the row's denominator
  const int cnt_raw = cand_count[grow_c];
  const int j_raw = cand_j[grow_c * a.slots + slot];
  const float x_raw = cand_x[grow_c * a.slots + slot];
  const float nmr_i = a.nmr[grow_c];
  const int cnt = row_ok ? min(cnt_raw, a.slots) : 0;
  const bool live = slot < cnt;
  float rs;
  if (dense) rs = a.sums_ready ? a.rsum[grow_c] : group_sum(share(a.rowB + (long)b_c * a.splits * a.Lp + i_c, a.splits, a.Lp));
  else rs = list_sum(__builtin_amdgcn_exp2f(__builtin_fmaf(x_raw, a.k, nmr_i)), j_raw, live, a.slots, base, wave_max_int(cnt));
  bool keep = false;
  int j = live ? j_raw : 0x7fffffff;
  float conf = 0.f, colbest = 0.f;
  if (dense && a.sums_ready) {
    // Dense samples whose denominators k_reduce_sums folded: FOUR entries of the row per pass.  Everything an entry needs
    // is two levels of loads - its column's stabiliser, denominator and list (lane s: the s-th entry of that list), then
    // the stabiliser and denominator of the ROW of every list entry, requested by the lane that holds it - and the four
    // entries' loads of a level are in flight together: a row without a peak holds up to `slots` candidates, and one
    // entry per pass with its two dependent round trips was 19 of this kernel's 21 us at 16 slots.
    for (int t0 = 0; t0 < a.slots; t0 += 4) {
      if (!__any(t0 < cnt)) break;                            // wave-uniform
      bool act[4];
      int jt[4], ccnt[4], ci[4];
      float xt[4], nmc[4], cs[4], cx[4], nm2_l[4], rs2_l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = t0 + q;
        act[q] = t < cnt;                                     // (a row's lanes agree)
        jt[q] = act[q] ? __shfl(j_raw, base + (t & (a.slots - 1))) : 0;
        xt[q] = __shfl(x_raw, base + (t & (a.slots - 1)));
        const long gcol = (long)b_c * a.Sp + jt[q];
        nmc[q] = a.nmc[gcol];
        ccnt[q] = ccand_count[gcol];
        ci[q] = ccand_i[gcol * a.slots + slot];
        cx[q] = ccand_x[gcol * a.slots + slot];
        cs[q] = a.csum[gcol];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ccnt[q] = act[q] ? min(ccnt[q], a.slots) : 0;
        const long g2 = (long)b_c * a.Lp + ((act[q] && slot < ccnt[q]) ? ci[q] : i_c);
        nm2_l[q] = a.nmr[g2];
        rs2_l[q] = a.rsum[g2];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ct = act[q] ? entry_conf(xt[q], a.k, nmr_i, rs, nmc[q], cs[q]) : 0.f;
        float cb = 0.f;
        const int cmax = wave_max_int(ccnt[q]);
        for (int e = 0; e < cmax; ++e) {                      // the column's entries: this entry among them
          const bool ea = e < ccnt[q];
          const int i2 = ea ? __shfl(ci[q], base + e) : i_c;
          const float x2 = __shfl(cx[q], base + e);
          const float nm2 = __shfl(nm2_l[q], base + e), rs2 = __shfl(rs2_l[q], base + e);
          const float c2 = entry_conf(x2, a.k, nm2, rs2, nmc[q], cs[q]);
          if (ea && i2 != i) cb = fmaxf(cb, c2);              // another row's entry of this column
          if (ea && i2 == i) cb = fmaxf(cb, ct);
        }
        if (act[q] && slot == t0 + q) { conf = ct; colbest = cb; }
      }
    }
  } else
  for (int t = 0; t < a.slots; ++t) {                         // entry t of every row of the wave that has one
    if (!__any(t < cnt)) break;                               // wave-uniform
    const bool act = t < cnt;                                 // (a row's lanes agree)
    const int jt = act ? __shfl(j_raw, base + t) : 0;
    const float xt = __shfl(x_raw, base + t);
    const long gcol = (long)b_c * a.Sp + jt;
    // round trip 2: the column's denominator, stabiliser, and its list - lane s the s-th entry
    const float nmc = a.nmc[gcol];
    const int ccnt = act ? min(ccand_count[gcol], a.slots) : 0;
    const int ci = ccand_i[gcol * a.slots + slot];
    const float cx = ccand_x[gcol * a.slots + slot];
    float cs;
    if (dense) cs = a.sums_ready ? a.csum[gcol] : group_sum(share(a.colB + (long)b_c * a.panels * a.Sp + jt, a.panels, a.Sp));
    else cs = list_sum(__builtin_amdgcn_exp2f(__builtin_fmaf(cx, a.k, nmc)), ci, slot < ccnt, a.slots, base, wave_max_int(ccnt));
    // (an entry that is negligible for its column - not in that column's list - has conf < 2^-32)
    const bool col_sig = dense || __builtin_fmaf(xt, a.k, nmc) > -kSkipLog2;
    const float ct = (act && col_sig) ? entry_conf(xt, a.k, nmr_i, rs, nmc, cs) : 0.f;
    float cb = 0.f;
    const int cmax = wave_max_int(ccnt);
    for (int e = 0; e < cmax; ++e) {                          // the column's entries: this entry among them
      const bool ea = e < ccnt;
      const int i2 = ea ? __shfl(ci, base + e) : i_c;
      const float x2 = __shfl(cx, base + e);
      const bool other = ea && i2 != i;                       // another row's entry (rare): that row's denominator
      if (__any(other)) {                                     // wave-uniform
        const int i2c = other ? i2 : i_c;
        const float nm2 = a.nmr[(long)b_c * a.Lp + i2c];
        const float rs2 = row_denominator(i2c, nm2);
        const float c2 = entry_conf(x2, a.k, nm2, rs2, nmc, cs);
        if (other && (dense || __builtin_fmaf(x2, a.k, nm2) > -kSkipLog2)) cb = fmaxf(cb, c2);
      }
      if (ea && i2 == i) cb = fmaxf(cb, ct);
    }
    if (act && slot == t) { conf = ct; colbest = cb; }
  }
  float rowbest = conf;
  for (int m = 1; m < a.slots; m <<= 1) rowbest = fmaxf(rowbest, __shfl_xor(rowbest, m));
#ifdef FM_DIAG_CLOCK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SEL_STAMP(2)
#endif
  if (live)
    keep = conf > a.thr && conf == rowbest && conf == colbest &&
           interior(i, a.h0c, a.w0c, a.border) && interior(j, a.h1c, a.w1c, a.border);
  // rank among the row's kept entries by ascending j (torch.where order, :109)
  const int kj = keep ? j : 0x7fffffff;
  int rank = 0, nkeep = 0;
  for (int s = 0; s < a.slots; ++s) {
    const int oj = __shfl(kj, base + s);
    rank += (oj < kj) ? 1 : 0;
    nkeep += (oj != 0x7fffffff) ? 1 : 0;
  }
  const int q = threadIdx.x / a.slots;                 // row within the workgroup (< 64: slots >= 4)
  if (keep) { s_kj[q * a.slots + rank] = j; s_kc[q * a.slots + rank] = conf; }
  if (slot == 0) rowcnt[q] = nkeep;
  // matches of this workgroup's rows -> published at once (value and flag in one word)
  int tot = (slot == 0) ? nkeep : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
  if (lane == 0) sm[threadIdx.x >> 6] = tot;
  __syncthreads();
  const int total = sm[0] + sm[1] + sm[2] + sm[3];
  if (threadIdx.x == 0)
    __hip_atomic_store(&a.blocktot[blk], total | (int)0x40000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  SEL_STAMP(3)
  // exclusive prefix of the totals of the blocks before this one (fixed order of integer adds)
  int pre = 0;
  int polls = 0;                 // ONE budget per thread for all of its predecessors: the worst-case wait is 2^22 polls
  for (int k = threadIdx.x; k < blk; k += 256) {
    int v;
    do {
      v = __hip_atomic_load(&a.blocktot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(v & 0x40000000)) {
        if (++polls > (1 << 22)) { atomicOr(&a.scal->flags, (unsigned)FM_INT_LOOKBACK_TIMEOUT); v = 0x40000000; }
        else __builtin_amdgcn_s_sleep(1);
      }
    } while (!(v & 0x40000000));
    pre += v & 0x3fffffff;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) pre += __shfl_xor(pre, m);
  __syncthreads();
  if (lane == 0) sm[threadIdx.x >> 6] = pre;
  __syncthreads();
  pre = sm[0] + sm[1] + sm[2] + sm[3];
  const int rows_per_block = 256 / a.slots;
  if (threadIdx.x < 64) {       // wave 0: inclusive scan of the row counts
    const int cntq = threadIdx.x < rows_per_block ? rowcnt[threadIdx.x] : 0;
    int incl = cntq;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (threadIdx.x < rows_per_block) rowoff[threadIdx.x] = pre + incl - cntq;
    // informational: did every sample go to the dense sum kernel?  (the caller's cue for FM_MODE_FLAT; last block only)
    bool all_dense = false;
    if (blk == a.nblk - 1 && a.dense_enabled) {
      bool mine = true;
      for (int bb = threadIdx.x; bb < a.N; bb += 64) mine = mine && a.dense_cnt[bb] > 0;
      all_dense = __all(mine);
    }
    if (blk == a.nblk - 1 && threadIdx.x == 63) {
      const int run = pre + incl;
      a.d_count[0] = run;
      // without the exact-screening pass an overflow of the sum kernels' candidate slots is final
      // (other workgroups set bits with device-scope atomics until they exit: read at device scope, behind the look-back)
      unsigned fl = __hip_atomic_load(&a.scal->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!a.exact && (fl & (unsigned)FM_INT_SCREEN_OVERFLOW)) fl |= (unsigned)FM_DEV_CANDIDATES;
      if (fl & (unsigned)FM_INT_LOOKBACK_TIMEOUT) fl |= (unsigned)FM_DEV_INTERNAL;
      a.d_count[1] = (int)((fl & (15u | (unsigned)FM_DEV_INTERNAL | (unsigned)FM_DEV_STEP) & ~(unsigned)FM_DEV_CAPACITY) |
                           (run > a.cap ? (unsigned)FM_DEV_CAPACITY : 0u) | (all_dense ? (unsigned)FM_DEV_ALL_DENSE : 0u));
    }
  }
  __syncthreads();
  SEL_STAMP(4)
#ifdef FM_DIAG_CLOCK
#define SEL_DIAG_OUT                                                                                  \
  if (threadIdx.x == 0) {                                                                             \
    SEL_STAMP(5)                                                                                      \
    for (int q2 = 0; q2 < 6; ++q2) a.diag[blk * 8 + q2] = (float)(dg[q2] & 0xffffff);                 \
  }
#else
#define SEL_DIAG_OUT
#endif
  if (b >= a.N || slot >= rowcnt[q]) { SEL_DIAG_OUT return; }
  const long o = (long)rowoff[q] + slot;
  if (o >= a.cap) { SEL_DIAG_OUT return; }
  const int jj = s_kj[q * a.slots + slot];
  a.b_ids[o] = b; a.i_ids[o] = i; a.j_ids[o] = jj;
  // cell -> match maps: with exact ties the largest match index keeps the cell, every other tied match
  // is listed once (whoever loses the atomicMax, now or when it is displaced later, is the one listed)
  if (a.cell_maps) {
    const int me = (int)o + 1;
    int old = atomicMax(&a.cell0[grow], me);
    int loser = old > me ? me : old;
    if (loser > 0) { const int p = atomicAdd(&a.ties0[0], 1); if (p < kTieCap) a.ties0[1 + p] = loser - 1; }
    old = atomicMax(&a.cell1[(long)b * a.Sp + jj], me);
    loser = old > me ? me : old;
    if (loser > 0) { const int p = atomicAdd(&a.ties1[0], 1); if (p < kTieCap) a.ties1[1 + p] = loser - 1; }
  }
  a.mconf[o] = s_kc[q * a.slots + slot];
  // coarse_matching_new.py:126-134: (x, y) = (id % w, id // w) * scale [* scale{0,1}[b]]
  float s0x = a.scale_px, s0y = a.scale_px, s1x = a.scale_px, s1y = a.scale_px;
  if (a.scale0) { s0x = a.scale_px * a.scale0[b * 2]; s0y = a.scale_px * a.scale0[b * 2 + 1]; }
  if (a.scale1) { s1x = a.scale_px * a.scale1[b * 2]; s1y = a.scale_px * a.scale1[b * 2 + 1]; }
  a.k0[o * 2] = (float)(i % a.w0c) * s0x; a.k0[o * 2 + 1] = (float)(i / a.w0c) * s0y;
  a.k1[o * 2] = (float)(jj % a.w1c) * s1x; a.k1[o * 2 + 1] = (float)(jj / a.w1c) * s1y;
#ifdef FM_DIAG_CLOCK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SEL_DIAG_OUT
}

hipError_t launch_select(const CoarseWs& w, char* base, int h0c, int w0c,
                         int h1c, int w1c, float inv_ct, float thr, int border, float scale_px,
                         const float* scale0, const float* scale1, int64_t* b_ids, int64_t* i_ids,
                         int64_t* j_ids, float* k0, float* k1, float* mconf, int cap, int32_t* d_count,
                         int mode, hipStream_t st, const MapCopyJob* job) {
  SelArgs a;
  a.exact = (mode & FM_MODE_EXACT_SCREENING) ? 1 : 0;
  a.dense_enabled = (mode & (FM_MODE_DENSE | FM_MODE_EXACT_SCREENING)) ? 1 : 0;
  a.cell_maps = (mode & FM_MODE_NO_CELL_MAPS) ? 0 : 1;
  a.nmr = (const float*)(base + w.nmr); a.nmc = (const float*)(base + w.nmc);
  a.rowB = (const float*)(base + w.rowB); a.colB = (const float*)(base + w.colB);
  a.cand_count = (const int*)(base + w.cand_count); a.cand_j = (const int*)(base + w.cand_j);
  a.cand_x = (const float*)(base + w.cand_x);
  a.ccand_count = (const int*)(base + w.ccand_count); a.ccand_i = (const int*)(base + w.ccand_i);
  a.ccand_x = (const float*)(base + w.ccand_x);
  a.cand_count_b = (const int*)(base + w.cand_count_b); a.cand_j_b = (const int*)(base + w.cand_j_b);
  a.cand_x_b = (const float*)(base + w.cand_x_b);
  a.ccand_count_b = (const int*)(base + w.ccand_count_b); a.ccand_i_b = (const int*)(base + w.ccand_i_b);
  a.ccand_x_b = (const float*)(base + w.ccand_x_b);
  a.dense_cnt = (const int*)(base + w.dense_cnt);
  a.rsum = (const float*)(base + w.rsum); a.csum = (const float*)(base + w.csum);
  a.sums_ready = (mode & (FM_MODE_EXACT_SCREENING | FM_MODE_STATS)) ? 1 : 0;     // (the caller ORs FM_MODE_STATS in whenever k_reduce_sums ran)
  a.blocktot = (int*)(base + w.blocktot);
  a.scal = (Scalars*)(base + w.scalars);
  a.N = w.N; a.L = w.L; a.S = w.S; a.C = w.C; a.Lp = w.Lp; a.Sp = w.Sp; a.splits = w.splits; a.splits_s = w.splits_s; a.panels = w.panels;
  a.slots = w.slots; a.h0c = h0c; a.w0c = w0c; a.h1c = h1c; a.w1c = w1c; a.border = border;
  a.k = inv_ct * kLog2e; a.thr = thr; a.scale_px = scale_px; a.scale0 = scale0; a.scale1 = scale1;
  a.b_ids = b_ids; a.i_ids = i_ids; a.j_ids = j_ids; a.k0 = k0; a.k1 = k1; a.mconf = mconf;
  a.cap = cap; a.d_count = d_count;
  a.cell0 = (int*)(base + w.cell0); a.cell1 = (int*)(base + w.cell1);
  a.ties0 = (int*)(base + w.ties0); a.ties1 = (int*)(base + w.ties1);
  const int blocks = (int)(((long)w.N * w.Lp * w.slots + 255) / 256);
  a.nblk = blocks;
  a.diag = (float*)(base + w.rowB);       // (diagnostic builds run on a full-size workspace)
  a.job = MapCopyJob{nullptr, nullptr, 0, 0, 0};
  int extra = 0;
  if (job && job->src) {
    // one workgroup per (sample, row, 64-pixel piece) up to four per compute unit, the rest in grid strides
    a.job = *job;
    const long units = (long)((job->Wf + 63) / 64) * job->Hf * job->N;
    extra = (int)(units < 1024 ? units : 1024);
  }
  hipLaunchKernelGGL(k_select, dim3(blocks + extra), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace fm
