// Coarse stage, the SCREENING kernels: which entries of the L x S product matter, and their exact values.
//
// Reproduces network/utils/coarse_matching_new.py:64-68 (correlation + dual softmax) for every entry whose
// term is not negligible.  After the max pass (k_max_i8: row / column maxima and the maximum of every
// 32 x 32 unit of the int8 screening product x~, error bounded by fm_device.h) each entry s_ij falls in one of
// three classes:
//   * negligible : k x~_ij + margin lies more than 2^32 below BOTH its row's and its column's stabiliser: it adds
//                  < 2^-32 to sums that are >= e^-2E ~ 1 (<= S 2^-32 ~ 1e-6 relative in total, inside the 1e-5
//                  parity bar) and cannot be a candidate.  Never touched again.  With dual-softmax-trained
//                  (peaked) descriptors that is all but ~1 entry per row.
//   * significant, few per row / column : the exact float32 dot product of the two descriptors is recomputed from the
//                  caller's rows (16 lanes per entry, 16 channels per lane, fixed order) and the entry (index, x) is
//                  appended to its ROW's list when exp2(k x - m^_i) > 2^-32 and to its COLUMN's list when
//                  exp2(k x - c^_j) > 2^-32.  k_select forms sum_j exp2(k x - m^_i), sum_i exp2(k x - c^_j) and conf
//                  from these lists (in index order: deterministic) - the same number in numerator and denominator of
//                  conf, as in the reference's softmax.
//   * significant, many (flat similarity: untrained network, repetitive texture, a partner that is missing) : a unit with
//                  more than kMaxExact significant entries, or a row / column with more than cand_slots, flags its
//                  SAMPLE for the dense sum kernel (k_dense: float32-equivalent hi/lo product on the matrix cores
//                  for all 1024 entries of every live unit), which redoes that sample when the call runs with
//                  FM_MODE_DENSE; without it the call reports FM_E_DENSE (fm_coarse_match_auto then adds the dense part).
//                  A sample is handled by ONE of the two kernels: their float32 products agree to ~1e-7 but not bit
//                  for bit, and coarse_matching_new.py:105-106 keeps exactly tied entries (conf == row max == column
//                  max), so identical descriptors must see one arithmetic.
//
// Kernels (round 5): k_thresh (one thread per row / column: stabilisers, integer thresholds, per-block maxima of them)
// and k_screen_rows (ONE independent wave per (32-row block, <= 64 column units): live units only, B fragments straight
// from global memory into registers, no barrier, 4 KiB of LDS per workgroup) - see the block comment in front of them
// for the one-round-trip, 128-KiB-of-LDS workgroup kernel of round 4 they replace and why.  k_stab serves FM_MODE_FLAT.
#include <stdlib.h>
#include <string.h>

#include "fm_device.h"

namespace fm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMaxExact = 24;         // significant entries of a unit resolved by exact dot products; more -> dense kernel

struct ScreenArgs {
  const signed char* q0; const signed char* q1;       // int8 screening planes (k_prep_split)
  const void* src0; const void* src1; int c_in, in_dtype;   // the caller's descriptors [N,L,c_in] / [N,S,c_in]
  const unsigned* rowmax_u; const unsigned* colmax_u;  // max pass: q_encode'd maxima of the integer screening product
  const float* sigimg;                                 // [N][2] quantisation step of image 0 / image 1
  const float* imgstat;                                // [N][8] {largest L1 norm, clipped mass, |x|} of image 0, of image 1 (max pass)
  const float* l1_0; const float* l1_1;                // L1 norms
  const float4* bstat0; const float4* bstat1;          // per 32-row block: {largest L1 norm, largest clipped L1 mass, ..}
  const float* umax;                                   // unit maxima of the integer screening product
  float* nmr; float* nmc; float* emarg;                // written here: stabilisers, pair margin
  int* dense_cnt; Scalars* scal;                       // [N] units per sample left to the dense kernel
  int* rcount; int* rlist_j; float* rlist_x;           // significant entries per row: columns, exact dot products
  int* ccount; int* clist_i; float* clist_x;           // ... per column: rows, exact dot products
  int L, S, Lp, Sp, slots, dense_enabled;
  int allow_dead;                                      // 0: every row / column keeps its stabiliser and its full sum (conf_matrix)
  float k, lt, inv_ct, cpad;
};

__device__ __forceinline__ int xcd_remap_s(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov_s(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xf, BANK, false));
}
// maximum over the 32 lanes that share lane >> 5 (every lane of the half ends with it)
__device__ __forceinline__ float half_reduce32_max(float v) {
  v = fmaxf(v, dpp_mov_s<0xB1, 0xf>(v, v));                                                       // lane ^ 1
  v = fmaxf(v, dpp_mov_s<0x4E, 0xf>(v, v));                                                       // lane ^ 2
  { float t = dpp_mov_s<0x104, 0x5>(v, v); t = dpp_mov_s<0x114, 0xA>(t, v); v = fmaxf(v, t); }   // lane ^ 4
  v = fmaxf(v, dpp_mov_s<0x128, 0xf>(v, v));                                                      // lane ^ 8
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = fmaxf(p, q); }
  return v;
}
// sum over the 16 lanes of a DPP row, in a fixed order (every lane of the row ends with the same bits)
__device__ __forceinline__ float row_sum16(float v) {
  v = v + dpp_mov_s<0xB1, 0xf>(v, v);            // quad_perm [1,0,3,2]
  v = v + dpp_mov_s<0x4E, 0xf>(v, v);            // quad_perm [2,3,0,1]
  v = v + dpp_mov_s<0x141, 0xf>(v, v);           // row_half_mirror
  v = v + dpp_mov_s<0x140, 0xf>(v, v);           // row_mirror
  return v;
}

// Integer significance threshold: an entry with integer screening product q can matter for a row / column whose
// -stabiliser*log2e is nm iff  kss q + emu + nm > -kSkipLog2  <=>  q > (-kSkipLog2 - emu - nm) / kss.
// floor() - 1 absorbs the float roundings of the quotient (a lower threshold only lets more entries through).
__device__ __forceinline__ int sig_threshold(float nm, float emu, float inv_kss) {
  const float t = floorf((-kSkipLog2 - emu - nm) * inv_kss) - 1.f;
  return (int)fminf(fmaxf(t, -1.0e9f), 1.0e9f);       // (NaN -> -1e9: everything significant)
}

// FM_MODE_FLAT: what the screening kernel does besides screening, for a call whose samples all go to the dense sum
// kernel - the stabilisers of every row and column (the SAME expressions as in k_screen: lower bounds from the int8 max
// pass, dead rows / columns at -inf), the pair margin and the range checks, the inverse scale of the float16 planes
// k_prep_split<C, true> wrote, and the dense flags.  grid (chunks of 256 lines, N, 2): z = 0 rows, 1 columns.
__global__ __launch_bounds__(256) void k_stab(ScreenArgs a, float* f16inv) {
  const int side = blockIdx.z, b = blockIdx.y;
  const int len = side ? a.S : a.L, lenp = side ? a.Sp : a.Lp;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float sig0 = a.sigimg[b * 2], sig1 = a.sigimg[b * 2 + 1];
  const float* ist = a.imgstat + (long)b * 8;
  const float l1A_max = ist[0], clipA = ist[1], infA = ist[2], l1B_max = ist[3], clipB = ist[4], infB = ist[5];
  const float ss = sig0 * sig1;
  if (blockIdx.x == 0 && side == 0 && threadIdx.x == 0) {
    const float emarg = margin_log2(q8_margin_raw(sig0, l1A_max, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
    a.emarg[b] = emarg;
    const bool clipped = clipA > 0.f || clipB > 0.f;
    // float16 planes: scale from the int8 step (k_prep_split<C, true>); the image's true maximum must fit
    const float sc0 = f16_plane_scale(127.f * sig0), sc1 = f16_plane_scale(127.f * sig1);
    f16inv[b] = (1.0f / sc0) * (1.0f / sc1);
    if (!(l1A_max < INFINITY) || !(l1B_max < INFINITY)) atomicOr(&a.scal->flags, (unsigned)FM_DEV_RANGE);
    else if (!(emarg < 60.f)) atomicOr(&a.scal->flags, (unsigned)(clipped ? FM_DEV_STEP : FM_DEV_RANGE));
    else if (!(infA * sc0 < 65504.f) || !(infB * sc1 < 65504.f)) atomicOr(&a.scal->flags, (unsigned)FM_DEV_STEP);
    a.dense_cnt[b] = 1;
    a.scal->dense_units = 1;
  }
  if (idx >= lenp) return;
  const long g = (long)b * lenp + idx;
  float nm = -INFINITY;                      // padded lines
  if (idx < len) {
    const float ln2 = 0.69314718f;
    const float l1 = (side ? a.l1_1 : a.l1_0)[g];
    const unsigned mx = (side ? a.colmax_u : a.rowmax_u)[g];
    const bool dead = a.allow_dead && 2.002f * l1 * (side ? infA : infB) * a.inv_ct + 1e-3f <
                                          (a.lt + __builtin_log2f((float)(side ? a.L : a.S))) * ln2;
    const float mraw = side ? q8_margin_raw(sig0, l1A_max, clipA, sig1, l1, clipB, a.cpad)
                            : q8_margin_raw(sig0, l1, clipA, sig1, l1B_max, clipB, a.cpad);
    nm = dead ? -INFINITY : neg_stabiliser_log2(ss * q_decode(mx), mraw, a.inv_ct);
  }
  (side ? a.nmc : a.nmr)[g] = nm;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_thresh + k_screen_rows (round 5).
//
// Round 4's k_screen was built for ONE pair per launch and one stream: a workgroup = 8 row blocks x <= 16 units, everything
// in one memory round trip, the int8 B fragments of ALL units of the range by LDS-DMA whether alive or not (64-128 KiB of
// LDS per workgroup), one barrier - 361 workgroups that each live ~7 us: 13 us per 640x480 pair.  Two things were wrong
// with that shape, both measured in round 5:
//   * at a batch of 64 pairs it is 23 104 workgroups in 45 rounds of two per compute unit (LDS), every one the same
//     latency chain: 523 us against ~90 us of bytes;
//   * on the bench's FOUR streams a workgroup that parks 128 KiB of LDS and 8 waves for 7 us keeps the other pairs' max
//     pass (64 KiB of LDS per workgroup) off its compute unit: with the kernels below - 5 us SLOWER alone at one pair -
//     the four-stream rate went from 25.4 k to 28.0 k pairs/s.
// What matters is that nothing couples waves to each other and that a wave holds few resources:
//   * k_thresh (one thread per row / column): stabilisers, integer significance thresholds, the largest stabiliser of
//     every 32-row block and 32-column unit, the pair margin and the range checks - what every k_screen workgroup
//     derived for its panel and range, computed once.  (Folding it into every wave of k_screen_rows - no launch, a
//     bound per unit from the unit's smallest column maximum - was tried: 25.3 us against 23.6 us alone, 26.9 k against
//     27.5 k pairs/s: the redundant column work is on every wave's critical path.)
//   * k_screen_rows: ONE WAVE per (row block, <= 64 units) item, no barrier, no LDS-DMA: the wave loads its A fragments
//     and thresholds, derives its live mask from the unit maxima, and walks the LIVE units only - B fragments straight
//     from global memory into registers (8 KiB per unit, the next unit's in flight while this one is screened: 16 KiB
//     per wave, 12 waves per compute unit keep its 64 B/clk busy), 8 MFMAs, the integer screening, the exact float32 dot
//     products and list reservations.  Items are ordered so that an XCD works on one or two samples at a time (their
//     int8 planes, 2.4 MiB, stay in its L2).  64 pairs: 530 -> 270 us (sweep 110 us, exact phase 150 us: 630 MB of
//     descriptor rows from HBM through the compute units' ~23 GB/s each); one 1024x1024 pair 75 -> 32 us.
// ---------------------------------------------------------------------------------------------------------------------
struct RowsExtra {
  int* thr_r; int* thr_c; float* wmaxb; float* cmaxu;
  int nchunks, chunk_units, items;       // chunk_units <= 64 units per item (one ballot covers a chunk)
};

// grid (chunks of 256 lines, N, 2): z = 0 rows, 1 columns; a 256-thread block = 8 row blocks / units of 32 lines
__global__ __launch_bounds__(256) void k_thresh(ScreenArgs a, RowsExtra x) {
  const int side = blockIdx.z, b = blockIdx.y;
  const int len = side ? a.S : a.L, lenp = side ? a.Sp : a.Lp;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const float sig0 = a.sigimg[b * 2], sig1 = a.sigimg[b * 2 + 1];
  const float* ist = a.imgstat + (long)b * 8;
  const float l1A_max = ist[0], clipA = ist[1], infA = ist[2], l1B_max = ist[3], clipB = ist[4], infB = ist[5];
  const float ss = sig0 * sig1;
  const float kss = a.k * ss;
  const bool screen_ok = kss > 1e-30f && kss < 1e30f;
  const float inv_kss = screen_ok ? 1.0f / kss : 0.f;
  if (blockIdx.x == 0 && side == 0 && threadIdx.x == 0) {      // (the same checks as k_screen's first workgroup)
    const float emarg = margin_log2(q8_margin_raw(sig0, l1A_max, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
    a.emarg[b] = emarg;
    const bool clipped = clipA > 0.f || clipB > 0.f;
    const float emarg0 = margin_log2(q8_margin_raw(sig0, l1A_max, 0.f, sig1, l1B_max, 0.f, a.cpad), a.inv_ct);
    if (!(l1A_max < INFINITY) || !(l1B_max < INFINITY)) atomicOr(&a.scal->flags, (unsigned)FM_DEV_RANGE);
    else if (!(emarg < 60.f)) atomicOr(&a.scal->flags, (unsigned)(clipped ? FM_DEV_STEP : FM_DEV_RANGE));
    else if (clipped && emarg > 2.0f * emarg0 + 1.0f) atomicOr(&a.scal->flags, (unsigned)FM_DEV_STEP);
  }
  if (idx >= lenp) return;                 // (lenp is a multiple of 64: whole waves leave)
  const long g = (long)b * lenp + idx;
  const int blk = idx >> 5;                // row block / unit of this line
  const int nblk = lenp / 32;
  float nm = -INFINITY;                    // padded lines: no stabiliser, nothing significant on their account
  float emu = 0.f;
  if (idx < len) {
    const float ln2 = 0.69314718f;
    const float l1 = (side ? a.l1_1 : a.l1_0)[g];
    const unsigned mx = (side ? a.colmax_u : a.rowmax_u)[g];
    const float bl1 = (side ? a.bstat1 : a.bstat0)[(long)b * nblk + blk].x;       // largest L1 norm of the line's block
    const bool dead = a.allow_dead && 2.002f * l1 * (side ? infA : infB) * a.inv_ct + 1e-3f <
                                          (a.lt + __builtin_log2f((float)(side ? a.L : a.S))) * ln2;
    const float mraw = side ? q8_margin_raw(sig0, l1A_max, clipA, sig1, l1, clipB, a.cpad)
                            : q8_margin_raw(sig0, l1, clipA, sig1, l1B_max, clipB, a.cpad);
    nm = dead ? -INFINITY : neg_stabiliser_log2(ss * q_decode(mx), mraw, a.inv_ct);
    // margin of any entry of this line's block against the other image's largest L1 norm
    emu = margin_log2(side ? q8_margin_raw(sig0, l1A_max, clipA, sig1, bl1, clipB, a.cpad)
                           : q8_margin_raw(sig0, bl1, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
  }
  (side ? a.nmc : a.nmr)[g] = nm;
  (side ? x.thr_c : x.thr_r)[g] = idx < len ? sig_threshold(nm, emu, inv_kss) : 0x3fffffff;
  const float bm = half_reduce32_max(nm);
  if ((lane & 31) == 0) (side ? x.cmaxu : x.wmaxb)[(long)b * nblk + blk] = bm;
}

template <int C>
__global__ __launch_bounds__(256, 3) void k_screen_rows(ScreenArgs a, RowsExtra x) {
  constexpr int KS8 = C / 32;
  constexpr int LIST = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  __shared__ int s_list[4][LIST];

  // item = (sample, chunk, group of 4 row blocks) -> the 4 waves of a workgroup take 4 consecutive row blocks of one
  // chunk: their live units overlap (L1 / L2 hits on the B fragments); through the bijective XCD remap one XCD's share
  // is a contiguous range of items, i.e. one or two samples at a time
  const int nrb = a.Lp / 32, nunits = a.Sp / 32;
  const int item = xcd_remap_s(blockIdx.x, gridDim.x) * 4 + wv;
  if (item >= x.items) return;                      // (wave-uniform; the kernel has no barrier)
  const int per_sample = nrb * x.nchunks;
  const int b = item / per_sample;
  int kk = item - b * per_sample;
  const int chunk = kk / nrb;
  const int rb = kk - chunk * nrb;
  const int wrow0 = rb * 32;
  if (wrow0 >= a.L) return;                         // nothing but padding rows
  const int u0 = chunk * x.chunk_units;
  const int U = min(x.chunk_units, nunits - u0);

  // ---- one round trip: steps and statistics, thresholds of this lane's 16 rows, unit maxima / largest column
  // stabilisers / largest L1 norms of the chunk's units (lane = unit), the A fragments ----
  const float sig0 = a.sigimg[b * 2], sig1 = a.sigimg[b * 2 + 1];
  const float* ist = a.imgstat + (long)b * 8;
  const float clipA = ist[1], clipB = ist[4];
  const float bl1A = a.bstat0[(long)b * nrb + rb].x;
  const float wmax = x.wmaxb[(long)b * nrb + rb];
  const int ul_c = min(lane, U - 1);
  const float um = a.umax[((long)b * nrb + rb) * nunits + u0 + ul_c];
  const float cm = x.cmaxu[(long)b * nunits + u0 + ul_c];
  const float bl1B = a.bstat1[(long)b * nunits + u0 + ul_c].x;
  int trr[16];
  {
    const int* tp = x.thr_r + (long)b * a.Lp + wrow0 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 t4 = *reinterpret_cast<const int4*>(tp + 8 * q);
      trr[4 * q] = t4.x; trr[4 * q + 1] = t4.y; trr[4 * q + 2] = t4.z; trr[4 * q + 3] = t4.w;
    }
  }
  v4i aq[KS8];
  {
    const signed char* src = a.q0 + (((long)b * a.Lp + wrow0) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) aq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
  }
  const float ss = sig0 * sig1;
  const float kss = a.k * ss;
  const bool screen_ok = kss > 1e-30f && kss < 1e30f;

  // ---- live units of this row block (the same bound as k_screen) ----
  unsigned long long live;
  {
    const float emu = margin_log2(q8_margin_raw(sig0, bl1A, clipA, sig1, bl1B, clipB, a.cpad), a.inv_ct);
    const float top = __builtin_fmaf(um, kss, emu);
    const bool hot = lane < U && (!screen_ok || !((top + wmax < -kSkipLog2) && (top + cm < -kSkipLog2)));
    live = __ballot(hot);
  }
  // A unit with more significant entries than the exact phase resolves (flat similarity) sends its whole SAMPLE to the
  // dense sum kernel, and from then on every list of the sample is dead weight: a wave stops at its first such unit, and
  // does not start when the sample is flagged already (a plain, possibly stale load: only an optimisation).
  const bool flat = !screen_ok || a.dense_cnt[b] != 0;
  int nd_units = flat ? 1 : 0;
  int nlist = 0;
#ifdef FM_ABL_ROWS          // ablation builds (tools/): 1 = no sweep, 2 = no exact phase
  if (FM_ABL_ROWS & 1) live = 0;
#endif
  if (!flat && live) {
    const bool row_edge = (wrow0 + 32 > a.L);
    const signed char* q1b = a.q1 + ((long)b * nunits + u0) * KS8 * 1024 + lane * 16;
    const int* tcb = x.thr_c + (long)b * a.Sp + u0 * 32 + r;
    auto load_unit = [&](int ul, v4i (&bq)[KS8], int& tcl) {
      const signed char* src = q1b + (long)ul * KS8 * 1024;
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) bq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
      tcl = tcb[ul * 32];
    };
    auto screen_unit = [&](int ul, const v4i (&bq)[KS8], int tcl) {
      const int ucol0 = (u0 + ul) * 32;
      v16i acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0;
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[ks], bq[ks], acc, 0, 0, 0);
      if (row_edge || ucol0 + 32 > a.S) {
        const bool cok = ucol0 + r < a.S;
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (!cok || wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h >= a.L) acc[g] = kQMasked;
      }
      unsigned bm = 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) bm |= (acc[g] > min(trr[g], tcl)) ? (1u << g) : 0u;
      unsigned long long hitl = __ballot(bm != 0);
      if (!hitl) return;
      // (count first: a flat unit has hundreds of hits, and walking them one by one is a scalar loop)
      int cnt = 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) cnt += __builtin_popcountll(__ballot((bm >> g) & 1u));
      if (cnt > kMaxExact || nlist + cnt > LIST) { ++nd_units; return; }
      while (hitl) {
        const int l = __builtin_ctzll(hitl);
        hitl &= hitl - 1;
        unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)bm, l);
        while (bits) {
          const int g = __builtin_ctz(bits);
          bits &= bits - 1;
          const int rl = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
          if (lane == 0) s_list[wv][nlist] = (ul << 10) | (rl << 5) | (l & 31);
          ++nlist;
        }
      }
    };
    // the live units two at a time: the next unit's fragments are in flight while this one is screened
    v4i bA[KS8], bB[KS8];
    int tcA = 0, tcB = 0;
    unsigned long long mask = live;
    int ulA = __builtin_ctzll(mask), ulB = -1;
    mask &= mask - 1;
    load_unit(ulA, bA, tcA);
    while (true) {
      ulB = -1;
      if (mask) { ulB = __builtin_ctzll(mask); mask &= mask - 1; load_unit(ulB, bB, tcB); }
      screen_unit(ulA, bA, tcA);
      if (ulB < 0 || nd_units) break;
      ulA = -1;
      if (mask) { ulA = __builtin_ctzll(mask); mask &= mask - 1; load_unit(ulA, bA, tcA); }
      screen_unit(ulB, bB, tcB);
      if (ulA < 0 || nd_units) break;
    }
    if (nd_units) nlist = 0;           // the sample goes to the dense kernel: its lists are not read
  }

  // ---- the parked entries: exact float32 dot products (as in k_screen; stabilisers from k_thresh's arrays) ----
  int overflow = 0;
#ifdef FM_ABL_ROWS
  if (FM_ABL_ROWS & 2) nlist = 0;
#endif
  {
    constexpr int NPASS = 4;
    const int sub = lane >> 4, l16 = lane & 15;
    const int vpr = a.c_in >> 2;
    for (int e0 = 0; e0 < nlist; e0 += 4 * NPASS) {
      int myres = 0;
      float my_nmr = 0.f, my_nmc = 0.f;
      if (lane < 4 * NPASS && e0 + lane < nlist) {
        const int ky = s_list[wv][e0 + lane];
        const int row = wrow0 + ((ky >> 5) & 31), col = (u0 + ((ky >> 10) & 63)) * 32 + (ky & 31);
        const int pr = atomicAdd(&a.rcount[(long)b * a.Lp + row], 1);
        const int pc = atomicAdd(&a.ccount[(long)b * a.Sp + col], 1);
        my_nmr = a.nmr[(long)b * a.Lp + row];
        my_nmc = a.nmc[(long)b * a.Sp + col];
        myres = min(pr, 0xffff) | (min(pc, 0xffff) << 16);
      }
      int key[NPASS];
      float xs[NPASS];
#pragma unroll
      for (int h2 = 0; h2 < NPASS; h2 += 2) {
        float4 av[2][4], bv[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int p = h2 + pp;
          const int idx = e0 + 4 * p + sub;
          key[p] = s_list[wv][min(idx, nlist - 1)];
          const int row = wrow0 + ((key[p] >> 5) & 31), col = (u0 + ((key[p] >> 10) & 63)) * 32 + (key[p] & 31);
          const long ro = ((long)b * a.L + row) * a.c_in, co = ((long)b * a.S + col) * a.c_in;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int v4 = l16 + 16 * q;
            const bool in = v4 < vpr;
            const int vc = in ? v4 : 0;
            float4 ta, tb;
            if (a.in_dtype == FM_F32) {
              ta = reinterpret_cast<const float4*>((const float*)a.src0 + ro)[vc];
              tb = reinterpret_cast<const float4*>((const float*)a.src1 + co)[vc];
            } else {
              ta = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)a.src0 + ro)[vc], a.in_dtype);
              tb = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)a.src1 + co)[vc], a.in_dtype);
            }
            av[pp][q] = in ? ta : make_float4(0.f, 0.f, 0.f, 0.f);
            bv[pp][q] = tb;
          }
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          float sm = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sm = __builtin_fmaf(av[pp][q].x, bv[pp][q].x, sm);
            sm = __builtin_fmaf(av[pp][q].y, bv[pp][q].y, sm);
            sm = __builtin_fmaf(av[pp][q].z, bv[pp][q].z, sm);
            sm = __builtin_fmaf(av[pp][q].w, bv[pp][q].w, sm);
          }
          xs[h2 + pp] = row_sum16(sm);
        }
        if (e0 + 4 * (h2 + 2) >= nlist) break;
      }
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int idx = e0 + 4 * p + sub;
        if (e0 + 4 * p >= nlist) break;
        const int pos = __shfl(myres, 4 * p + sub);
        const float nmr_e = __shfl(my_nmr, 4 * p + sub), nmc_e = __shfl(my_nmc, 4 * p + sub);
        if (l16 == 0 && idx < nlist) {
          const int row = wrow0 + ((key[p] >> 5) & 31), col = (u0 + ((key[p] >> 10) & 63)) * 32 + (key[p] & 31);
          const float rr = __builtin_fmaf(xs[p], a.k, nmr_e);
          const float cc = __builtin_fmaf(xs[p], a.k, nmc_e);
          const int pr = pos & 0xffff, pc = (pos >> 16) & 0xffff;
          const long grow = (long)b * a.Lp + row, gcol = (long)b * a.Sp + col;
          if (pr < a.slots) { a.rlist_j[grow * a.slots + pr] = col; a.rlist_x[grow * a.slots + pr] = rr > -kSkipLog2 ? xs[p] : -INFINITY; }
          else overflow = 1;
          if (pc < a.slots) { a.clist_i[gcol * a.slots + pc] = row; a.clist_x[gcol * a.slots + pc] = cc > -kSkipLog2 ? xs[p] : -INFINITY; }
          else overflow = 1;
        }
      }
    }
  }
  nd_units += __builtin_popcountll(__ballot(overflow != 0));
  if (nd_units && lane == 0) {
    a.dense_cnt[b] = 1;
    a.scal->dense_units = 1;
    if (!a.dense_enabled) atomicOr(&a.scal->flags, (unsigned)FM_DEV_DENSE);
  }
}

static void fill_screen_stats(ScreenArgs& a, const CoarseWs& w, char* base, float inv_ct, float thr, int allow_dead) {
  a.rowmax_u = (const unsigned*)(base + w.rowmax_u); a.colmax_u = (const unsigned*)(base + w.colmax_u);
  a.sigimg = (const float*)(base + w.sigimg); a.imgstat = (const float*)(base + w.imgstat);
  a.l1_0 = (const float*)(base + w.l1_0); a.l1_1 = (const float*)(base + w.l1_1);
  a.bstat0 = (const float4*)(base + w.bstat0); a.bstat1 = (const float4*)(base + w.bstat1);
  a.umax = (const float*)(base + w.umax);
  a.nmr = (float*)(base + w.nmr); a.nmc = (float*)(base + w.nmc); a.emarg = (float*)(base + w.emarg);
  a.dense_cnt = (int*)(base + w.dense_cnt); a.scal = (Scalars*)(base + w.scalars);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp;
  a.slots = w.slots; a.allow_dead = allow_dead;
  a.k = inv_ct * kLog2e; a.lt = log2f(thr); a.inv_ct = inv_ct; a.cpad = (float)w.C;
}

hipError_t launch_stab(const CoarseWs& w, char* base, float inv_ct, float thr, int allow_dead, hipStream_t st) {
  ScreenArgs a;
  memset(&a, 0, sizeof(a));
  fill_screen_stats(a, w, base, inv_ct, thr, allow_dead);
  const int lenp = w.Lp > w.Sp ? w.Lp : w.Sp;
  hipLaunchKernelGGL(k_stab, dim3((lenp + 255) / 256, w.N, 2), dim3(256), 0, st, a, (float*)(base + w.f16inv));
  return hipGetLastError();
}

hipError_t launch_screen(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                         float inv_ct, float thr, int dense_enabled, int allow_dead, hipStream_t st) {
  ScreenArgs a;
  memset(&a, 0, sizeof(a));
  fill_screen_stats(a, w, base, inv_ct, thr, allow_dead);
  a.in_dtype = in_dtype;
  a.q0 = (const signed char*)(base + w.q0); a.q1 = (const signed char*)(base + w.q1);
  a.src0 = feat0; a.src1 = feat1; a.c_in = c_in;
  a.rcount = (int*)(base + w.cand_count); a.rlist_j = (int*)(base + w.cand_j); a.rlist_x = (float*)(base + w.cand_x);
  a.ccount = (int*)(base + w.ccand_count); a.clist_i = (int*)(base + w.ccand_i); a.clist_x = (float*)(base + w.ccand_x);
  a.dense_enabled = dense_enabled;
  RowsExtra x;
  x.thr_r = (int*)(base + w.thr_r); x.thr_c = (int*)(base + w.thr_c);
  x.wmaxb = (float*)(base + w.wmaxb); x.cmaxu = (float*)(base + w.cmaxu);
  x.chunk_units = w.units_s;
  x.nchunks = w.splits_s;
#ifdef FM_TUNE_ENV
  if (const char* ev = getenv("FM_ROWS_CHUNK")) { x.chunk_units = atoi(ev); x.nchunks = (w.Sp / 32 + x.chunk_units - 1) / x.chunk_units; }
#endif
  x.items = (int)((long)w.N * (w.Lp / 32) * x.nchunks);
  const int lenp = w.Lp > w.Sp ? w.Lp : w.Sp;
  hipLaunchKernelGGL(k_thresh, dim3((lenp + 255) / 256, w.N, 2), dim3(256), 0, st, a, x);
  const int blocks_r = (x.items + 3) / 4;
  switch (w.C) {
    case 64: hipLaunchKernelGGL(k_screen_rows<64>, dim3(blocks_r), dim3(256), 0, st, a, x); break;
    case 128: hipLaunchKernelGGL(k_screen_rows<128>, dim3(blocks_r), dim3(256), 0, st, a, x); break;
    case 256: hipLaunchKernelGGL(k_screen_rows<256>, dim3(blocks_r), dim3(256), 0, st, a, x); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace fm
