// Coarse stage, the SPARSE screening kernel: which entries of the L x S product matter, and their exact values.
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
//                  conf, as in the reference's softmax.  Round 4: the kernel no longer forms the sums itself (eight
//                  waves' accumulators in LDS, a fold, 6 + 19 partial-sum arrays for the assignment to add up again):
//                  a list of <= cand_slots (index, x) pairs per row and per column carries the same information.
//   * significant, many (flat similarity: untrained network, repetitive texture, a partner that is missing) : a unit with
//                  more than kMaxExact significant entries, or a row / column with more than cand_slots, flags its
//                  SAMPLE for the dense sum kernel (k_dense: float32-equivalent hi/lo product on the matrix cores
//                  for all 1024 entries of every live unit), which redoes that sample when the call runs with
//                  FM_MODE_DENSE; without it the call reports FM_E_DENSE and the caller repeats it with the flag.
//                  A sample is handled by ONE of the two kernels: their float32 products agree to ~1e-7 but not bit
//                  for bit, and coarse_matching_new.py:105-106 keeps exactly tied entries (conf == row max == column
//                  max), so identical descriptors must see one arithmetic.
//
// Structure: a workgroup = 8 waves = 8 row blocks (32 rows) x one range of <= 16 column units; ONE barrier.  Everything
// the decisions depend on - the images' steps and statistics (reduced once per sample by the max pass: no per-wave
// reductions over the block statistics), the rows' and the range's column maxima and L1 norms, the unit maxima of all
// eight row blocks, the wave's int8 A fragments - is requested at once (one memory round trip); each wave turns its 32
// rows and its share of the columns into stabilisers and integer thresholds (LDS).  Behind the first barrier every
// wave derives the live-unit masks of all eight row blocks itself (eight compares + ballots); the units that are alive
// for ANY row block of the panel are brought into LDS ONCE, by LDS-DMA, all of them in flight together (wave w the
// w-th 1 KiB piece of every unit), and behind the second barrier every wave runs its own row block's live units from
// there: ds_read_b128 of the B fragments, 8 v_mfma_i32_32x32x32_i8, the integer screening of the 32 x 32 accumulators.
// Round 4 measured why: with ~1 unit in 5 alive the workgroup's (row block, unit) pairs are ~40 x 16 KiB of operands
// through ONE compute unit's 64 B/clk vector-memory path - more bytes than the max pass moves for the whole product -
// and every unit's fragments were a dependent L2 round trip of 2-3k cycles under that load (sweep: 2.1k cycles per
// unit against 0.26k of matrix time).  Sharing a unit's fragments between the row blocks that need them halves the
// bytes and takes the round trips out of the sweep.  The parked entries are resolved four at a time, eight in flight.
// No LDS accumulators, no fold, no partial sums.
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
  float* diag;                                         // diagnostic build: stamp buffer
  int* rcount; int* rlist_j; float* rlist_x;           // significant entries per row: columns, exact dot products
  int* ccount; int* clist_i; float* clist_x;           // ... per column: rows, exact dot products
  int L, S, Lp, Sp, panels, splits, units_s, slots, pgroup, dense_enabled;
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
// butterfly over the 64 lanes (every lane ends with the same bits: each level adds / maxes disjoint pairs)
template <bool SUM>
__device__ __forceinline__ float wave_reduce64(float v) {
  auto op = [](float x, float y) { return SUM ? x + y : fmaxf(x, y); };
  v = op(v, dpp_mov_s<0xB1, 0xf>(v, v));                                                       // lane ^ 1
  v = op(v, dpp_mov_s<0x4E, 0xf>(v, v));                                                       // lane ^ 2
  { float t = dpp_mov_s<0x104, 0x5>(v, v); t = dpp_mov_s<0x114, 0xA>(t, v); v = op(v, t); }   // lane ^ 4
  v = op(v, dpp_mov_s<0x128, 0xf>(v, v));                                                      // lane ^ 8
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = op(p, q); }
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = op(p, q); }
  return v;
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

template <int C>
__global__ __launch_bounds__(512) void k_screen(ScreenArgs a) {
  constexpr int KS8 = C / 32;           // k-steps of v_mfma_i32_32x32x32_i8
  constexpr int LIST = 256;             // significant entries a wave can park (beyond that a unit goes to the dense kernel)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // workgroup order as in k_dense: sample, groups of a.pgroup panels, split-major inside a group, through the
  // bijective XCD remap - one XCD's share is a compact (panels x splits) block (speed only)
  int kk = xcd_remap_s(blockIdx.x, gridDim.x);
  const int per_sample = a.panels * a.splits;
  const int b = kk / per_sample;
  kk -= b * per_sample;
  const int gsz = a.pgroup * a.splits;
  const int pg = kk / gsz;
  kk -= pg * gsz;
  const int pcount = min(a.pgroup, a.panels - pg * a.pgroup);
  const int split = kk / pcount;
  const int panel = pg * a.pgroup + (kk - split * pcount);
  const int nunits = a.Sp / 32;
  const int u0 = split * a.units_s;
  const int U = max(0, min(a.units_s, nunits - u0));          // units of this workgroup's range (<= 64)
  const int rb = panel * 8 + wv;                              // this wave's row block
  const int wrow0 = rb * 32;
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps per phase of every wave (tools/diag_sparse.py)
  unsigned long long dg[8];
  dg[0] = __builtin_amdgcn_s_memtime();
#define DIAG_STAMP(i) dg[i] = __builtin_amdgcn_s_memtime();
#else
#define DIAG_STAMP(i)
#endif

  __shared__ __attribute__((aligned(16))) int s_tr[8][32];         // integer significance thresholds of the 8 waves' rows
  __shared__ float s_cmax[kScreenUnits];       // largest (least negative) column stabiliser of every unit of the range
  __shared__ float s_wmax[8];                  // ... row stabiliser of every row block
  __shared__ float s_nmr[8][32];               // -stabiliser*log2e of the 8 waves' rows
  __shared__ int s_list[8][LIST];              // (row block << 16) | (unit << 10) | (row in block << 5) | column in unit
  // dynamic LDS, sized for the range: column stabilisers [U*32], column thresholds [U*32], then the int8 B fragments
  // of the range's live units (KS8 KiB each)
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_nmc = s_dyn;
  int* s_tc = reinterpret_cast<int*>(s_dyn + a.units_s * 32);
  char* s_b = reinterpret_cast<char*>(s_dyn + 2 * a.units_s * 32);

  // ---- everything the decisions below depend on is requested at once (ONE memory round trip: every load below is
  // issued before the first use of any of them; indices are clamped instead of predicated so that no load sits behind
  // a branch): the images' steps and statistics, this wave's row statistics, the range's column statistics, the unit
  // maxima of all 8 row blocks and the wave's A fragments (lane (r,h): row r, k = h*C/2 + 8*ks + 0..7; one contiguous
  // 1 KiB block per k-step of the fragment-major plane of k_prep_split) ----
  constexpr int CSETS = 1;                              // column sets per thread: the range has <= 16 * 32 = 512 columns
  static_assert(kScreenUnits * 32 <= 512, "one column of the range per thread");
  const float sig0 = a.sigimg[b * 2], sig1 = a.sigimg[b * 2 + 1];
  const float* ist = a.imgstat + (long)b * 8;
  const float l1A_max = ist[0], clipA = ist[1], infA = ist[2], l1B_max = ist[3], clipB = ist[4], infB = ist[5];
  const int nb0 = a.Lp / 32;
  const float4* bs0 = a.bstat0 + (long)b * nb0;
  const float4* bs1 = a.bstat1 + (long)b * nunits;
  const long gi = (long)b * a.Lp + wrow0 + r;
  const unsigned rmax_u = a.rowmax_u[gi];
  const float rl1 = a.l1_0[gi];
  float bl1A8[8];                                                 // largest L1 norm of every row block of the panel
#pragma unroll
  for (int k = 0; k < 8; ++k) bl1A8[k] = bs0[panel * 8 + k].x;
  unsigned cmax_u[CSETS];
  float cl1[CSETS], cbl1[CSETS];
#pragma unroll
  for (int k = 0; k < CSETS; ++k) {
    const int c = min(tid + 512 * k, max(U * 32 - 1, 0));
    cmax_u[k] = a.colmax_u[(long)b * a.Sp + u0 * 32 + c];
    cl1[k] = a.l1_1[(long)b * a.Sp + u0 * 32 + c];
    cbl1[k] = bs1[min(u0 + (c >> 5), nunits - 1)].x;              // largest L1 norm of the column's unit
  }
  // lane u: unit u0 + u of every row block of the panel
  const int ul_c = min(lane, max(U - 1, 0));
  float um8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) um8[k] = a.umax[((long)b * nb0 + panel * 8 + k) * nunits + min(u0 + ul_c, nunits - 1)];
  const float bl1B_lane = bs1[min(u0 + ul_c, nunits - 1)].x;      // largest L1 norm of the unit's columns
  v4i aq[KS8];
  {
    const signed char* src = a.q0 + (((long)b * a.Lp + wrow0) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) aq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
  }
  // ... and the int8 B fragments of EVERY unit of the range: global -> LDS by LDS-DMA, unit n in slot n (KS8 KiB), piece
  // p = (n, ks) one 1 KiB instruction (the 64 lanes of an MFMA operand fragment), the waves take the pieces round
  // robin.  Unconditional: which units are alive is known one memory round trip later, and requesting only those then
  // was a second, dependent round trip of ~4k cycles for ~20 % fewer bytes.
  for (int p = wv; p < U * KS8; p += 8) {
    const signed char* src = a.q1 + (((long)b * nunits + u0) * KS8 + p) * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s_b + p * 1024), 16, 0, 0);
  }

  const float ss = sig0 * sig1;                    // integer screening product -> raw dot-product units
  const float kss = a.k * ss;                      // ... -> log2-domain similarity
  const bool screen_ok = kss > 1e-30f && kss < 1e30f;   // (an all-zero sample image: nothing to screen with)
  const float inv_kss = screen_ok ? 1.0f / kss : 0.f;
  DIAG_STAMP(1)
  if (panel == 0 && split == 0 && tid == 0) {
    const float emarg = margin_log2(q8_margin_raw(sig0, l1A_max, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
    a.emarg[b] = emarg;
    // a prep workgroup that met NaN/Inf/|x| >= 32768 reported +inf; descriptors so large that the screening margin
    // alone could overflow exp2 (similarities of several thousand) are out of range as well
    // ... unless the margin is what it is because the int8 step, estimated from a sample of the image's rows, clipped a
    // descriptor outside the sample (an outlier several times larger than the rest; a sample of textureless cells):
    // then the inputs are fine and the call is to be repeated with the exact step (FM_MODE_EXACT_STEP), as it is when
    // the clipped mass is more than half of the margin (every entry would look significant: the dense kernel's job for
    // no reason)
    const bool clipped = clipA > 0.f || clipB > 0.f;
    const float emarg0 = margin_log2(q8_margin_raw(sig0, l1A_max, 0.f, sig1, l1B_max, 0.f, a.cpad), a.inv_ct);
    if (!(l1A_max < INFINITY) || !(l1B_max < INFINITY)) atomicOr(&a.scal->flags, (unsigned)FM_DEV_RANGE);
    else if (!(emarg < 60.f)) atomicOr(&a.scal->flags, (unsigned)(clipped ? FM_DEV_STEP : FM_DEV_RANGE));
    else if (clipped && emarg > 2.0f * emarg0 + 1.0f) atomicOr(&a.scal->flags, (unsigned)FM_DEV_STEP);
  }

  // ---- stabilisers and integer thresholds: this wave's 32 rows, the workgroup's column range ----
  // lanes 0..31 (and their mirror 32..63): -stabiliser*log2e of row wrow0 + r
  // DEAD rows and columns.  Every similarity of row i lies in [-B, B] with B = ||a_i||_1 max|b| / (C T) (Hoelder), so its
  // softmax terms are all <= e^{2B} / S, and conf <= that: a row with e^{2B} / S < thr (a near-zero descriptor: a
  // textureless cell) cannot hold a match, whatever the other image looks like, and nobody ever reads its denominator.
  // Its entries then only matter for their COLUMNS: the row gets the stabiliser of a padding row (-inf: no entry is
  // significant on its account, never listed, the unit test ignores it).  Without this every entry of such a row is
  // within 2^32 of the row's (tiny) maximum, i.e. significant, and one textureless patch sends the whole sample to the
  // dense kernel - 24x the time.  Columns likewise.  (1.001, 1e-3: the float roundings of B.)
  const float ln2 = 0.69314718f;
  const bool dead_row = a.allow_dead && 2.002f * rl1 * infB * a.inv_ct + 1e-3f < (a.lt + __builtin_log2f((float)a.S)) * ln2;
  const float nm_lane = dead_row ? -INFINITY
                                 : neg_stabiliser_log2(ss * q_decode(rmax_u),
                                                       q8_margin_raw(sig0, rl1, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
  // margin of any entry of this row block (its largest L1 norm against the other image's largest)
  const float emu_rows = margin_log2(q8_margin_raw(sig0, bl1A8[wv], clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
  if (h == 0) {
    s_tr[wv][r] = sig_threshold(nm_lane, emu_rows, inv_kss);
    s_nmr[wv][r] = nm_lane;
    if (split == 0) a.nmr[gi] = nm_lane;
  }
  {   // padded rows (>= L) never contribute: keep them out of the wave's largest stabiliser
    const float wm = wave_reduce64<false>(wrow0 + r < a.L ? nm_lane : -INFINITY);
    if (lane == 0) s_wmax[wv] = wm;
  }
  // (a wave's 64 consecutive columns are two units: their largest stabilisers - padded and dead columns excluded - come
  // out of the same pass by a butterfly over each half of the wave)
#pragma unroll
  for (int k = 0; k < CSETS; ++k) {
    const int c = tid + 512 * k;
    float nm = -INFINITY;
    if (c < U * 32) {
      const long gj = (long)b * a.Sp + u0 * 32 + c;
      const bool dead_col = a.allow_dead && 2.002f * cl1[k] * infA * a.inv_ct + 1e-3f < (a.lt + __builtin_log2f((float)a.L)) * ln2;
      nm = dead_col ? -INFINITY
                    : neg_stabiliser_log2(ss * q_decode(cmax_u[k]),
                                          q8_margin_raw(sig0, l1A_max, clipA, sig1, cl1[k], clipB, a.cpad), a.inv_ct);
      s_nmc[c] = nm;
      s_tc[c] = sig_threshold(nm, margin_log2(q8_margin_raw(sig0, l1A_max, clipA, sig1, cbl1[k], clipB, a.cpad), a.inv_ct), inv_kss);
      if (panel == 0) a.nmc[gj] = nm;
    }
    if (64 * wv + 512 * k < U * 32) {                       // wave-uniform: this wave holds columns of the range
      const float v = half_reduce32_max(u0 * 32 + c < a.S ? nm : -INFINITY);
      const int ul = 2 * wv + 16 * k + h;
      if (r == 0 && ul < U) s_cmax[ul] = v;
    }
  }
  __syncthreads();                  // the one barrier: thresholds and stabilisers of the panel and the range are in LDS,
                                    // and so are the B fragments (the barrier's vmcnt(0) drains the LDS-DMA)

  // ---- which units of the panel's 8 row blocks are alive (same bound as the dense kernel's block-sparse skip); every
  // wave derives all eight masks itself: lane u = log2-domain bound of k * |screening product - exact product| over
  // unit u of row block k ----
  unsigned masks[8];
  int tot = 0;
  {
    const float cm = lane < U ? s_cmax[lane] : -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float emu = margin_log2(q8_margin_raw(sig0, bl1A8[k], clipA, sig1, bl1B_lane, clipB, a.cpad), a.inv_ct);
      const float top = __builtin_fmaf(um8[k], kss, emu);          // >= k * (exact product), log2 domain
      const bool hot = lane < U && (panel * 8 + k) * 32 < a.L &&
                       (!screen_ok || !((top + s_wmax[k] < -kSkipLog2) && (top + cm < -kSkipLog2)));
      masks[k] = (unsigned)__ballot(hot);
      tot += __builtin_popcount(masks[k]);
    }
  }
  // more than half of the block's units are alive (and more than a handful): flat similarity, a matrix-core job (a
  // tiny block - a tiny image - is cheap to sweep whatever is alive, and truly flat units still overflow kMaxExact below)
  const bool flat = (tot * 2 > 8 * U && tot >= 12) || !screen_ok;
  // units left to the dense kernel: a flat block is reported ONCE, by wave 0 (every wave reporting its own share was
  // 3800 atomics on two addresses per 640x480 pair: 40 us of a 57 us launch on flat data); otherwise per wave, rarely
  int nd_units = (flat && wv == 0) ? tot : 0;

  int nlist = 0;                 // parked significant entries (wave-uniform)
  DIAG_STAMP(2)
  DIAG_STAMP(3)
#ifdef FM_DIAG_CLOCK
  int diag_units = 0;
#endif
  if (!flat && masks[wv]) {
    int trr[16];                 // this lane's 16 row thresholds (rows 8q + 4h + 0..3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 t4 = *reinterpret_cast<const int4*>(&s_tr[wv][8 * q + 4 * h]);
      trr[4 * q] = t4.x; trr[4 * q + 1] = t4.y; trr[4 * q + 2] = t4.z; trr[4 * q + 3] = t4.w;
    }
    const bool row_edge = (wrow0 + 32 > a.L);
    unsigned mask = masks[wv];
    while (mask) {
      const int ul = __builtin_ctz(mask);
      mask &= mask - 1;
#ifdef FM_DIAG_CLOCK
      ++diag_units;
#endif
      const v4i* bsl = reinterpret_cast<const v4i*>(s_b + (long)ul * KS8 * 1024) + lane;
      const int ucol0 = (u0 + ul) * 32;
      v16i acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0;
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[ks], bsl[ks * 64], acc, 0, 0, 0);
      if (row_edge || ucol0 + 32 > a.S) {              // padded rows (>= L) / columns (>= S) never count
        const bool cok = ucol0 + r < a.S;
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (!cok || wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h >= a.L) acc[g] = kQMasked;
      }
      // significance: the integer product beats the row's or the column's threshold (k x~ + margin within 2^32 of
      // that stabiliser).  Every lane collects the hits of its 16 registers in a bit mask - four vector instructions per
      // register, no branch: the first version's ballot + branch per register were 16 TAKEN branches per unit, ~1k of
      // the unit's 1.8k cycles - and ONE ballot tells whether anybody has any; the (rare) significant entries are parked
      // at once, in (lane, register) order, and taken back if the unit turns out to have too many of them (flat
      // similarity: the dense kernel's job).
      const int tcl = s_tc[ul * 32 + r];
      const int nlist0 = nlist;
      unsigned bm = 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) bm |= (acc[g] > min(trr[g], tcl)) ? (1u << g) : 0u;
      unsigned long long hitl = __ballot(bm != 0);
      while (hitl) {                   // wave-uniform, one or two lanes
        const int l = __builtin_ctzll(hitl);
        hitl &= hitl - 1;
        unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)bm, l);
        while (bits) {
          const int g = __builtin_ctz(bits);
          bits &= bits - 1;
          const int rl = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
          if (lane == 0 && nlist < LIST) s_list[wv][nlist] = (wv << 16) | (ul << 10) | (rl << 5) | (l & 31);
          ++nlist;
        }
      }
      if (nlist - nlist0 > kMaxExact || nlist > LIST) { nlist = nlist0; ++nd_units; }
    }
  }

  // ---- the parked entries: exact float32 dot products of the caller's descriptors.  16 lanes per entry (16 channels
  // per lane: four 16-byte loads from each of the two rows, 256 contiguous bytes per load instruction and entry), four
  // entries per pass, two passes (sixteen loads per lane) in flight; the sum of an entry is formed in a fixed order
  // (a chain of 16 fused multiply-adds per lane, then the 16 lanes by row_sum16).  The entry goes to its row's list
  // when its row term exceeds 2^-32, to its column's list when its column term does. ----
  DIAG_STAMP(4)
  int overflow = 0;              // lists that ran out of slots (per lane)
  {
    constexpr int NPASS = 4;     // 16 entries (32 loads per lane) in flight: the usual wave has < 10
    const int sub = lane >> 4, l16 = lane & 15;
    const int vpr = a.c_in >> 2;              // 4-channel vectors per descriptor
    for (int e0 = 0; e0 < nlist; e0 += 4 * NPASS) {
      // A place in its row's and in its column's list is reserved for every entry of the batch FIRST (lane t: entry
      // e0 + t): the returning atomics travel together with the loads of the exact rows below (behind the dot products
      // they were one more dependent round trip of ~3.5k cycles).  An entry that turns out to be negligible for one of
      // the two keeps its place there as an empty one (x = -inf: a zero term, never a candidate).
      int myres = 0;
      if (lane < 4 * NPASS && e0 + lane < nlist) {
        const int ky = s_list[wv][e0 + lane];
        const int rp = (ky >> 16) * 32 + ((ky >> 5) & 31), col = (u0 + ((ky >> 10) & 63)) * 32 + (ky & 31);
        const int pr = atomicAdd(&a.rcount[(long)b * a.Lp + panel * kPanelRows + rp], 1);
        const int pc = atomicAdd(&a.ccount[(long)b * a.Sp + col], 1);
        myres = min(pr, 0xffff) | (min(pc, 0xffff) << 16);
      }
      int key[NPASS];
      float xs[NPASS];
#pragma unroll
      for (int h2 = 0; h2 < NPASS; h2 += 2) {         // two passes' loads are issued together, then their sums
        float4 av[2][4], bv[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int p = h2 + pp;
          const int idx = e0 + 4 * p + sub;
          key[p] = s_list[wv][min(idx, nlist - 1)];             // (clamped: a repeated entry is computed and dropped)
          const int rp = (key[p] >> 16) * 32 + ((key[p] >> 5) & 31), col = (u0 + ((key[p] >> 10) & 63)) * 32 + (key[p] & 31);
          const long ro = ((long)b * a.L + panel * kPanelRows + rp) * a.c_in, co = ((long)b * a.S + col) * a.c_in;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int v4 = l16 + 16 * q;                        // this lane's q-th vector of the row
            const bool in = v4 < vpr;
            const int vc = in ? v4 : 0;                         // (clamped + select: no load behind a branch)
            float4 ta, tb;
            if (a.in_dtype == FM_F32) {
              ta = reinterpret_cast<const float4*>((const float*)a.src0 + ro)[vc];
              tb = reinterpret_cast<const float4*>((const float*)a.src1 + co)[vc];
            } else {       // float16 / bfloat16 rows: exact in float32, products of two halves are exact too
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
          xs[h2 + pp] = row_sum16(sm);           // the exact float32 dot product, same bits in the entry's 16 lanes
        }
        if (e0 + 4 * (h2 + 2) >= nlist) break;   // (wave-uniform: no entry in the remaining passes)
      }
      // the batch's entries go to the places reserved for them
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int idx = e0 + 4 * p + sub;
        if (e0 + 4 * p >= nlist) break;          // (wave-uniform)
        const int pos = __shfl(myres, 4 * p + sub);
        if (l16 == 0 && idx < nlist) {
          const int kk2 = key[p] >> 16, ul = (key[p] >> 10) & 63, rl = (key[p] >> 5) & 31, cl = key[p] & 31;
          const int rp = kk2 * 32 + rl, col = (u0 + ul) * 32 + cl;
          const float rr = __builtin_fmaf(xs[p], a.k, s_nmr[kk2][rl]);
          const float cc = __builtin_fmaf(xs[p], a.k, s_nmc[ul * 32 + cl]);
          const int pr = pos & 0xffff, pc = (pos >> 16) & 0xffff;
          const long grow = (long)b * a.Lp + panel * kPanelRows + rp, gcol = (long)b * a.Sp + col;
          if (pr < a.slots) { a.rlist_j[grow * a.slots + pr] = col; a.rlist_x[grow * a.slots + pr] = rr > -kSkipLog2 ? xs[p] : -INFINITY; }
          else overflow = 1;
          if (pc < a.slots) { a.clist_i[gcol * a.slots + pc] = panel * kPanelRows + rp; a.clist_x[gcol * a.slots + pc] = cc > -kSkipLog2 ? xs[p] : -INFINITY; }
          else overflow = 1;
        }
      }
    }
  }
  DIAG_STAMP(5)
  // ---- this wave's share of the sample's dense-unit count: units with too many significant entries, rows / columns
  // with more of them than slots (one atomic per wave, usually none) ----
  // (both words are read as "non-zero" only - by the plane kernel, the dense sum kernel, the assignment: plain stores
  // of 1, idempotent.  As atomic adds of the unit counts they were ~1000 read-modify-writes on two addresses when every
  // workgroup of a flat 640x480 pair reported: the kernel took 20.4 us on flat data against 14.6 us on peaked data.)
  nd_units += __builtin_popcountll(__ballot(overflow != 0));
  if (nd_units && lane == 0) {
    a.dense_cnt[b] = 1;
    a.scal->dense_units = 1;
    if (!a.dense_enabled) atomicOr(&a.scal->flags, (unsigned)FM_DEV_DENSE);   // nobody will redo this sample
  }
#ifdef FM_DIAG_CLOCK
  DIAG_STAMP(6)
  if (lane < 8) {        // stamps go to the dense kernel's row partials (unused while nothing is flagged)
    float vv = 0.f;
    const float vals[8] = {(float)(dg[1] - dg[0]), (float)(dg[2] - dg[1]), (float)(dg[3] - dg[2]), (float)(dg[4] - dg[3]),
                           (float)(dg[5] - dg[4]), (float)(dg[6] - dg[5]), (float)diag_units, (float)nlist};
#pragma unroll
    for (int q = 0; q < 8; ++q) vv = lane == q ? vals[q] : vv;
    a.diag[((long)blockIdx.x * 8 + wv) * 8 + lane] = vv;
  }
#endif
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
// BATCHED screening (round 5): k_thresh + k_screen_rows.
//
// k_screen above is built for ONE pair per launch: everything in one memory round trip, all B fragments of a range by
// LDS-DMA whether alive or not, one barrier - 361 workgroups that each live ~7 us.  At a batch of 64 pairs that shape
// is 23 104 workgroups in 45 rounds of two per compute unit (LDS), every one the same latency chain: 523 us against
// ~90 us of bytes.  With thousands of row blocks in a launch the latency of one wave does not matter - what matters is
// that nothing couples waves to each other and that enough of them are resident:
//   * k_thresh (one thread per row / column): stabilisers, integer significance thresholds, the largest stabiliser of
//     every 32-row block and 32-column unit, the pair margin and the range checks - what every k_screen workgroup
//     derives for its panel and range, computed once.
//   * k_screen_rows: ONE WAVE per (row block, <= 64 units) item, no barrier, no LDS-DMA: the wave loads its A fragments
//     and thresholds, derives its live mask from the unit maxima, and walks the LIVE units only - B fragments straight
//     from global memory into registers (8 KiB per unit, the next unit's in flight while this one is screened: 16 KiB
//     per wave, 12 waves per compute unit keep its 64 B/clk busy), 8 MFMAs, the same integer screening, the same exact
//     float32 dot products and list reservations as k_screen.  Items are ordered so that an XCD works on one or two
//     samples at a time (their int8 planes, 2.4 MiB, stay in its L2).
// The two forms produce the same lists up to the order of appends (k_select sums in index order) and flag the same
// kind of units for the dense kernel; which of them runs is a function of the shapes only.
// ---------------------------------------------------------------------------------------------------------------------
struct RowsExtra {
  int* thr_r; int* thr_c; float* wmaxb; float* cmaxu;
  int nchunks, items;
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
  const int u0 = chunk * 64;
  const int U = min(64, nunits - u0);

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
  const int tot = __builtin_popcountll(live);
  // more than half of the chunk's units alive AND more than 32 rows with one peak each can light up: flat similarity,
  // the dense kernel's job.  (32 peaks that happen to fall into 12 of a short last chunk's 22 units are not: on a batch
  // of 64 peaked pairs ~40 of the 9728 row blocks look like that.  A short chunk of truly flat data is swept and its
  // units overflow kMaxExact instead.)
  const bool flat = (tot * 2 > U && tot > 32) || !screen_ok;
  int nd_units = flat ? tot : 0;
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
      const int nlist0 = nlist;
      unsigned bm = 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) bm |= (acc[g] > min(trr[g], tcl)) ? (1u << g) : 0u;
      unsigned long long hitl = __ballot(bm != 0);
      while (hitl) {
        const int l = __builtin_ctzll(hitl);
        hitl &= hitl - 1;
        unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)bm, l);
        while (bits) {
          const int g = __builtin_ctz(bits);
          bits &= bits - 1;
          const int rl = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
          if (lane == 0 && nlist < LIST) s_list[wv][nlist] = (ul << 10) | (rl << 5) | (l & 31);
          ++nlist;
        }
      }
      if (nlist - nlist0 > kMaxExact || nlist > LIST) { nlist = nlist0; ++nd_units; }
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
      if (ulB < 0) break;
      ulA = -1;
      if (mask) { ulA = __builtin_ctzll(mask); mask &= mask - 1; load_unit(ulA, bA, tcA); }
      screen_unit(ulB, bB, tcB);
      if (ulA < 0) break;
    }
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
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels;
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
  a.in_dtype = in_dtype;
  a.q0 = (const signed char*)(base + w.q0); a.q1 = (const signed char*)(base + w.q1);
  a.src0 = feat0; a.src1 = feat1; a.c_in = c_in;
  a.rowmax_u = (const unsigned*)(base + w.rowmax_u); a.colmax_u = (const unsigned*)(base + w.colmax_u);
  a.sigimg = (const float*)(base + w.sigimg); a.imgstat = (const float*)(base + w.imgstat);
  a.l1_0 = (const float*)(base + w.l1_0); a.l1_1 = (const float*)(base + w.l1_1);
  a.bstat0 = (const float4*)(base + w.bstat0); a.bstat1 = (const float4*)(base + w.bstat1);
  a.umax = (const float*)(base + w.umax);
  a.nmr = (float*)(base + w.nmr); a.nmc = (float*)(base + w.nmc); a.emarg = (float*)(base + w.emarg);
  a.dense_cnt = (int*)(base + w.dense_cnt); a.scal = (Scalars*)(base + w.scalars);
  a.diag = (float*)(base + w.rowB);      // (diagnostic builds run on a full-size workspace)
  a.rcount = (int*)(base + w.cand_count); a.rlist_j = (int*)(base + w.cand_j); a.rlist_x = (float*)(base + w.cand_x);
  a.ccount = (int*)(base + w.ccand_count); a.clist_i = (int*)(base + w.ccand_i); a.clist_x = (float*)(base + w.ccand_x);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.splits = w.splits_s; a.units_s = w.units_s;
  a.slots = w.slots; a.dense_enabled = dense_enabled; a.allow_dead = allow_dead;
  {
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.units_s * 32) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
  }
  a.k = inv_ct * kLog2e; a.lt = log2f(thr); a.inv_ct = inv_ct; a.cpad = (float)w.C;
  // The batched form (k_thresh + k_screen_rows: one independent wave per (row block, 64 units)) when the launch holds
  // enough row blocks to fill the chip with such waves several times over; below that the one-round-trip kernel.
  {
    const long nrb = (long)w.N * (w.Lp / 32);
    int min_rb = kRowsFormMinRowBlocks;
#ifdef FM_TUNE_ENV
    if (const char* ev = getenv("FM_ROWS_MIN_RB")) min_rb = atoi(ev);
#endif
    if (nrb >= min_rb) {
      RowsExtra x;
      x.thr_r = (int*)(base + w.thr_r); x.thr_c = (int*)(base + w.thr_c);
      x.wmaxb = (float*)(base + w.wmaxb); x.cmaxu = (float*)(base + w.cmaxu);
      x.nchunks = (w.Sp / 32 + 63) / 64;
      x.items = (int)(nrb * x.nchunks);
      a.allow_dead = allow_dead;
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
  }
  const int blocks = w.N * a.splits * w.panels;
  const int smem = a.units_s * (32 * 2 * 4 + w.C * 32);     // column stabilisers + thresholds + B fragments of the range
  hipError_t e = hipSuccess;
#define FM_SCREEN_CASE(CC)                                                                   \
  case CC: {                                                                                 \
    static unsigned long long lds_set = 0;                                                   \
    e = ensure_dynamic_lds(&k_screen<CC>, kScreenUnits * (32 * 2 * 4 + CC * 32), &lds_set);   \
    if (e != hipSuccess) return e;                                                           \
    hipLaunchKernelGGL(k_screen<CC>, dim3(blocks), dim3(512), smem, st, a);                  \
    break;                                                                                   \
  }
  switch (w.C) {
    FM_SCREEN_CASE(64)
    FM_SCREEN_CASE(128)
    FM_SCREEN_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_SCREEN_CASE
  return hipGetLastError();
}

}  // namespace fm
