// Coarse stage, the SPARSE sum kernel: dual-softmax denominators and the candidate list from the units
// that matter, without barriers inside the sweep.
//
// Reproduces network/utils/coarse_matching_new.py:64-68 (correlation + dual softmax) for every entry whose
// term is not negligible.  After the max pass (k_max_i8: row / column maxima and the maximum of every
// 32 x 32 unit of the int8 screening product x~, error bounded by fm_device.h) each entry s_ij falls in one of
// three classes:
//   * negligible : k x~_ij + margin lies more than 2^32 below BOTH its row's and its column's stabiliser: it adds
//                  < 2^-32 to sums that are >= e^-2E ~ 1 (<= S 2^-32 ~ 1e-6 relative in total, inside the 1e-5
//                  parity bar) and cannot be a candidate.  Never touched again.  With dual-softmax-trained
//                  (peaked) descriptors that is all but ~1 entry per row.
//   * significant, few per unit : the exact float32 dot product of the two descriptors is recomputed from the
//                  caller's rows (one wave, 4 channels per lane, fixed butterfly) and enters
//                  sum_j exp2(k x - m^_i), sum_i exp2(k x - c^_j) and, when both terms pass log2 thr, the
//                  candidate list - the same number in numerator and denominator of conf, as in the reference.
//   * significant, many per unit (flat similarity: untrained network, repetitive texture) : the unit's SAMPLE is
//                  flagged for the dense sum kernel (k_corr<C,1>: float32-equivalent hi/lo product on the matrix
//                  cores for all 1024 entries of every live unit), which redoes that sample when the call runs with
//                  FM_MODE_DENSE; without it the call reports FM_E_DENSE and the caller repeats it with the flag.
//                  A sample is handled by ONE of the two kernels: their float32 products agree to ~1e-7 but not bit for bit, and coarse_matching_new.py:105-106 keeps exactly
//                  tied entries (conf == row max == column max), so identical descriptors must see one arithmetic.
//
// Structure: a workgroup = 8 INDEPENDENT waves = 8 row blocks (32 rows) x one range of <= 16 column units; no
// LDS tile ring and no barrier in the sweep - with ~1 unit in 5 alive, lock-stepping 8 waves through shared
// tiles cost the old sum sweep 19k of its 55k cycles in barrier waits at 640x480.  A wave keeps its 32 rows as
// int8 A-fragments in 32 VGPRs, decides from the unit maxima which of its units are alive, and for each of them
// loads the int8 B-fragments straight into registers (8 x 1 KiB contiguous blocks of the fragment-major plane at
// C = 256 - the sweep is bound by these L2 -> CU bytes, which is why it runs on the int8 planes - with the next
// unit's in flight behind the current chain), runs 8 v_mfma_i32_32x32x32_i8, screens the
// 32 x 32 accumulators against the stabilisers and resolves the significant entries exactly.  All sums are
// formed in a fixed order: results are bitwise reproducible.
#include "fm_device.h"

namespace fm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMaxExact = 24;         // significant entries of a unit resolved by exact dot products; more -> dense kernel
constexpr int kSparseQueue = 64;      // candidates a wave parks in LDS (one per lane at the hand-over)

struct SparseArgs {
  const signed char* q0; const signed char* q1;       // int8 screening planes (k_prep_split)
  const void* src0; const void* src1; int c_in, in_dtype;   // the caller's descriptors [N,L,c_in] / [N,S,c_in]
  const unsigned* rowmax_u; const unsigned* colmax_u;  // max pass: q_encode'd maxima of the integer screening product
  const float* sigimg;                                 // [N][2] quantisation step of image 0 / image 1
  const float* l1_0; const float* l1_1;                // L1 norms
  const float4* bstat0; const float4* bstat1;          // per 32-row block: {largest L1 norm, largest clipped L1 mass, ..}
  const float* umax;                                   // unit maxima of the integer screening product
  float* nmr; float* nmc; float* emarg;                // written here: stabilisers, pair margin
  float* rowS; float* colS;                            // partial sums [N][splits][Lp], [N][panels][Sp]
  int* dense_cnt; Scalars* scal;                       // [N] units per sample left to the dense kernel
  float* diag;                                         // diagnostic build: stamp buffer
  int* cand_count; int* cand_j; float* cand_x;         // candidates per row: columns, exact dot products
  int* ccand_count; int* ccand_i; float* ccand_x;      // the same candidates per column: rows, exact dot products
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

// Integer significance threshold: an entry with integer screening product q can matter for a row / column whose
// -stabiliser*log2e is nm iff  kss q + emu + nm > -kSkipLog2  <=>  q > (-kSkipLog2 - emu - nm) / kss.
// floor() - 1 absorbs the float roundings of the quotient (a lower threshold only lets more entries through).
__device__ __forceinline__ int sig_threshold(float nm, float emu, float inv_kss) {
  const float t = floorf((-kSkipLog2 - emu - nm) * inv_kss) - 1.f;
  return (int)fminf(fmaxf(t, -1.0e9f), 1.0e9f);       // (NaN -> -1e9: everything significant)
}

template <int C>
__global__ __launch_bounds__(512) void k_sum_sparse(SparseArgs a) {
  constexpr int KS8 = C / 32;           // k-steps of v_mfma_i32_32x32x32_i8
  constexpr int LIST = 256;             // significant entries a wave can park (beyond that a unit goes to the dense kernel)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // workgroup order as in k_corr: sample, groups of a.pgroup panels, split-major inside a group, through the
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
  __shared__ float s_cmax[kUnitsPerSplit];
  __shared__ int s_hot[8];
  // dynamic LDS, sized for the range: column stabilisers [U*32], column thresholds [U*32], then the 8 waves' column
  // accumulators [8][U*32]
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* s_nmc = s_dyn;
  int* s_tc = reinterpret_cast<int*>(s_dyn + a.units_s * 32);
  float* s_colacc = s_dyn + 2 * a.units_s * 32;
  const int cpitch = a.units_s * 32;
  __shared__ int s_list[8][LIST];              // (row block << 16) | (unit << 10) | (row in block << 5) | column in unit
  __shared__ int s_qkey[8][kSparseQueue];      // candidate queue: row inside the panel, column, dot product
  __shared__ int s_qcol[8][kSparseQueue];
  __shared__ float s_qx[8][kSparseQueue];
  __shared__ unsigned long long s_mask[8];     // live units of the 8 waves' row blocks
  __shared__ float s_nmr[8][32];               // -stabiliser*log2e of the 8 waves' rows
  __shared__ float s_rowacc[8 * kPanelRows];   // per wave: row sums of the entries it resolved, all 256 rows of the panel

  // ---- everything the decisions below depend on is requested at once (ONE memory round trip: every load below is
  // issued before the first use of any of them; indices are clamped instead of predicated so that no load sits behind
  // a branch): the images' steps, the block statistics, this wave's row statistics, the range's column statistics,
  // the unit maxima and the wave's A fragments (lane (r,h): row r, k = h*C/2 + 8*ks + 0..7; one contiguous 1 KiB
  // block per k-step of the fragment-major plane of k_prep_split) ----
  constexpr int CSETS = kUnitsPerSplit * 32 / 512;      // column sets per thread: the range has <= 2048 columns
  const float sig0 = a.sigimg[b * 2], sig1 = a.sigimg[b * 2 + 1];
  const int nb0 = a.Lp / 32;
  const float4* bs0 = a.bstat0 + (long)b * nb0;
  const float4* bs1 = a.bstat1 + (long)b * nunits;
  float4 st0[8], st1[8];           // lane: blocks lane + 64 q (a repeated block does not change a maximum)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    st0[q] = bs0[min(q * 64 + lane, nb0 - 1)];
    st1[q] = bs1[min(q * 64 + lane, nunits - 1)];
  }
  const long gi = (long)b * a.Lp + wrow0 + r;
  const unsigned rmax_u = a.rowmax_u[gi];
  const float rl1 = a.l1_0[gi];
  const float bl1A = bs0[rb].x;                                   // largest L1 norm of this wave's rows
  unsigned cmax_u[CSETS];
  float cl1[CSETS], cbl1[CSETS];
#pragma unroll
  for (int k = 0; k < CSETS; ++k) {
    const int c = min(tid + 512 * k, max(U * 32 - 1, 0));
    cmax_u[k] = a.colmax_u[(long)b * a.Sp + u0 * 32 + c];
    cl1[k] = a.l1_1[(long)b * a.Sp + u0 * 32 + c];
    cbl1[k] = bs1[min(u0 + (c >> 5), nunits - 1)].x;              // largest L1 norm of the column's unit
  }
  // lane u: unit u0 + u of this wave's row block
  const int ul_c = min(lane, max(U - 1, 0));
  float um = a.umax[((long)b * (a.Lp / 32) + rb) * nunits + min(u0 + ul_c, nunits - 1)];
  const float bl1B_lane = bs1[min(u0 + ul_c, nunits - 1)].x;      // largest L1 norm of the unit's columns
  v4i aq[KS8];
  {
    const signed char* src = a.q0 + (((long)b * a.Lp + wrow0) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) aq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
  }
  for (int c = lane; c < U * 32; c += 64) s_colacc[wv * cpitch + c] = 0.f;
  for (int c = lane; c < kPanelRows; c += 64) s_rowacc[wv * kPanelRows + c] = 0.f;

  float l1A_max = 0.f, l1B_max = 0.f, clipA = 0.f, clipB = 0.f;     // image maxima (every wave folds them itself)
  float infA = 0.f, infB = 0.f;                                     // largest |element| of image 0 / image 1
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    l1A_max = fmaxf(l1A_max, st0[q].x); clipA = fmaxf(clipA, st0[q].y); infA = fmaxf(infA, st0[q].z);
    l1B_max = fmaxf(l1B_max, st1[q].x); clipB = fmaxf(clipB, st1[q].y); infB = fmaxf(infB, st1[q].z);
  }
  for (int i = 512 + lane; i < nb0; i += 64) {
    const float4 v = bs0[i];
    l1A_max = fmaxf(l1A_max, v.x); clipA = fmaxf(clipA, v.y); infA = fmaxf(infA, v.z);
  }
  for (int i = 512 + lane; i < nunits; i += 64) {
    const float4 v = bs1[i];
    l1B_max = fmaxf(l1B_max, v.x); clipB = fmaxf(clipB, v.y); infB = fmaxf(infB, v.z);
  }
  l1A_max = wave_reduce64<false>(l1A_max);
  clipA = wave_reduce64<false>(clipA);
  l1B_max = wave_reduce64<false>(l1B_max);
  clipB = wave_reduce64<false>(clipB);
  infA = wave_reduce64<false>(infA);
  infB = wave_reduce64<false>(infB);
  if (lane >= U) um = -INFINITY;
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
  // significant on its account, no candidate, the unit test ignores it).  Without this every entry of such a row is
  // within 2^32 of the row's (tiny) maximum, i.e. significant, and one textureless patch sends the whole sample to the
  // dense kernel - 24x the time.  Columns likewise.  (1.001, 1e-3: the float roundings of B.)
  const float ln2 = 0.69314718f;
  const bool dead_row = a.allow_dead && 2.002f * rl1 * infB * a.inv_ct + 1e-3f < (a.lt + __builtin_log2f((float)a.S)) * ln2;
  const float nm_lane = dead_row ? -INFINITY
                                 : neg_stabiliser_log2(ss * q_decode(rmax_u),
                                                       q8_margin_raw(sig0, rl1, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
  // margin of any entry of this row block (its largest L1 norm against the other image's largest)
  const float emu_rows = margin_log2(q8_margin_raw(sig0, bl1A, clipA, sig1, l1B_max, clipB, a.cpad), a.inv_ct);
  if (h == 0) {
    s_tr[wv][r] = sig_threshold(nm_lane, emu_rows, inv_kss);
    s_nmr[wv][r] = nm_lane;
    if (split == 0) a.nmr[gi] = nm_lane;
  }
  // padded rows (>= L) never contribute: keep them out of the wave's largest stabiliser
  const float wmax_nmr = wave_reduce64<false>(wrow0 + r < a.L ? nm_lane : -INFINITY);
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
  __syncthreads();

  // ---- which of this wave's units are alive (same bound as the dense kernel's block-sparse skip) ----
  // lane u: log2-domain bound of k * |screening product - exact product| over unit u of this wave's row block
  const float emu_lane = margin_log2(q8_margin_raw(sig0, bl1A, clipA, sig1, bl1B_lane, clipB, a.cpad), a.inv_ct);
  bool hot = false;
  if (lane < U && wrow0 < a.L) {
    const float top = __builtin_fmaf(um, kss, emu_lane);          // >= k * (exact product), log2 domain
    hot = !screen_ok || !((top + wmax_nmr < -kSkipLog2) && (top + s_cmax[lane] < -kSkipLog2));
  }
  const unsigned long long own_mask = __ballot(hot);          // wave-uniform
  if (lane == 0) s_mask[wv] = own_mask;
  __syncthreads();
  // ---- the live units of the WHOLE workgroup, dealt out in equal shares ----
  // A wave's own row block holds between 0 and ~2.5x the average number of live units (each is ~1.7k cycles of
  // dependent loads + MFMAs + screening, and the workgroup ends with its slowest wave), so the (row block, unit)
  // pairs of the workgroup are put in one order - row block major - and wave w takes the w-th share of
  // ceil(total / 8).  A share is one run of that order: it lies in one or two (rarely more) row blocks; for a row
  // block that is not its own a wave loads that block's A fragments and thresholds.  Which wave handles what is a
  // function of the data only: all sums are still formed in a fixed order.
  unsigned long long masks[8];
  int tot = 0;
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const unsigned long long m = s_mask[w8];
    masks[w8] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    tot += __builtin_popcountll(masks[w8]);
  }
  // more than half of a block of >= 64 units is alive: flat similarity, a matrix-core job (a small block - a tiny
  // image, a one-unit split - is cheap to sweep whatever is alive, and truly flat units still overflow kMaxExact below)
  const bool flat = (tot * 2 > 8 * U && 8 * U >= 64) || !screen_ok;
  int nd_units = flat ? __builtin_popcountll(own_mask) : 0;      // units this wave leaves to the dense kernel

  int nlist = 0;                 // parked significant entries (wave-uniform)
  DIAG_STAMP(2)
#ifdef FM_DIAG_CLOCK
  int diag_units = 0;
  dg[3] = dg[2];
#endif
  if (!flat && tot) {
    const int chunk = (tot + 7) >> 3, lo = wv * chunk, hi = min(tot, lo + chunk);
    int cur_k = wv;                // row block whose A fragments sit in aq
    int trr[16];                   // this lane's 16 row thresholds of the current row block (rows 8q + 4h + 0..3)
    v4i bq0[KS8], bq1[KS8], bq2[KS8];
    int base = 0;
    for (int k = 0; k < 8 && base < hi; ++k) {
      const int ck = __builtin_popcountll(masks[k]);
      const int first = max(lo - base, 0), keep = min(hi, base + ck) - max(lo, base);
      base += ck;
      if (keep <= 0) continue;
      unsigned long long mask = masks[k];
      for (int d = 0; d < first; ++d) mask &= mask - 1;                       // drop the units of earlier shares
      { unsigned long long sub = 0, t2 = mask;                                // ... and of later ones
        for (int c = 0; c < keep; ++c) { sub |= t2 & (0 - t2); t2 &= t2 - 1; }
        mask = sub; }
#ifdef FM_DIAG_CLOCK
      diag_units += __builtin_popcountll(mask);
#endif
      const int wrow0k = (panel * 8 + k) * 32;
      const bool row_edge = (wrow0k + 32 > a.L);
      // B fragments of three units: while one feeds the MFMA chain the next two units' 16 KiB are in flight
      auto load_b = [&](v4i (&bq)[KS8], int ul) {
        const signed char* src = a.q1 + (((long)b * nunits + u0 + ul) * KS8 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < KS8; ++ks) bq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
      };
      // the unit after next (or `cur` again when there is none: the prefetch is UNCONDITIONAL - behind a branch hipcc
      // has to assume the shorter path at the join and waits vmcnt(0) in front of the MFMA chain, i.e. for the
      // prefetch it has just issued; measured 2.5k cycles per unit instead of ~1k)
      auto after_next = [](unsigned long long m, int cur) {
        const unsigned long long m2 = m & (m - 1);
        return m2 ? __builtin_ctzll(m2) : (m ? __builtin_ctzll(m) : cur);
      };
      {
        const int first_u = __builtin_ctzll(mask);
        load_b(bq0, first_u);
        const unsigned long long m1 = mask & (mask - 1);
        load_b(bq1, m1 ? __builtin_ctzll(m1) : first_u);
      }
      if (k != cur_k) {            // another wave's row block: its A fragments (the thresholds are in LDS)
        const signed char* src = a.q0 + (((long)b * a.Lp + wrow0k) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < KS8; ++ks) aq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
        cur_k = k;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 t4 = *reinterpret_cast<const int4*>(&s_tr[k][8 * q + 4 * h]);
        trr[4 * q] = t4.x; trr[4 * q + 1] = t4.y; trr[4 * q + 2] = t4.z; trr[4 * q + 3] = t4.w;
      }
      auto unit = [&](const v4i (&bq)[KS8], int ul) {
        const int ucol0 = (u0 + ul) * 32;
        v16i acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0;
#pragma unroll
        for (int ks = 0; ks < KS8; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[ks], bq[ks], acc, 0, 0, 0);
        if (row_edge || ucol0 + 32 > a.S) {              // padded rows (>= L) / columns (>= S) never count
          const bool cok = ucol0 + r < a.S;
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (!cok || wrow0k + (g & 3) + 8 * (g >> 2) + 4 * h >= a.L) acc[g] = kQMasked;
        }
        // significance: the integer product beats the row's or the column's threshold (k x~ + margin within 2^32 of
        // that stabiliser).  Two integer instructions per accumulator register write the wave's mask straight to
        // scalar registers; the (rare) significant entries are parked at once, in (register, lane) order, and taken
        // back if the unit turns out to have too many of them (flat similarity: the dense kernel's job).
        const int tcl = s_tc[ul * 32 + r];
        const int nlist0 = nlist;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          unsigned long long m = __ballot(acc[g] > min(trr[g], tcl));
          while (m) {                    // wave-uniform, usually not entered
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int rl = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
            if (lane == 0 && nlist < LIST) s_list[wv][nlist] = (k << 16) | (ul << 10) | (rl << 5) | (l & 31);
            ++nlist;
          }
        }
        if (nlist - nlist0 > kMaxExact || nlist > LIST) { nlist = nlist0; ++nd_units; }
      };
#ifdef FM_DIAG_CLOCK
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (dg[3] == dg[2]) { DIAG_STAMP(3) }
#endif
      while (mask) {
        const int ua = __builtin_ctzll(mask);
        mask &= mask - 1;                              // bq1 holds ctz(mask) now
        load_b(bq2, after_next(mask, ua));
        unit(bq0, ua);
        if (!mask) break;
        const int ub = __builtin_ctzll(mask);
        mask &= mask - 1;
        load_b(bq0, after_next(mask, ub));
        unit(bq1, ub);
        if (!mask) break;
        const int uc = __builtin_ctzll(mask);
        mask &= mask - 1;
        load_b(bq1, after_next(mask, uc));
        unit(bq2, uc);
      }
    }
  }

  // ---- the parked entries: exact float32 dot products of the caller's descriptors, eight entries (sixteen row
  // loads per lane) in flight; sums in list order (deterministic) ----
  int qn = 0;                    // parked candidates (wave-uniform)
  DIAG_STAMP(4)
  // a candidate goes to its row's AND its column's slot list (k_select takes the row best from the one, the column
  // best from the other, without a grid-wide pass in between); rp = row inside the panel
  auto record = [&](int rp, int col, float x) {
    const long grow = (long)b * a.Lp + panel * kPanelRows + rp, gcol = (long)b * a.Sp + col;
    const int pos = atomicAdd(&a.cand_count[grow], 1);
    const int cpos = atomicAdd(&a.ccand_count[gcol], 1);
    if (pos < a.slots) { a.cand_j[grow * a.slots + pos] = col; a.cand_x[grow * a.slots + pos] = x; }
    if (cpos < a.slots) { a.ccand_i[gcol * a.slots + cpos] = panel * kPanelRows + rp; a.ccand_x[gcol * a.slots + cpos] = x; }
    if (pos >= a.slots || cpos >= a.slots) atomicOr(&a.scal->flags, (unsigned)FM_INT_SCREEN_OVERFLOW);
  };
  constexpr int EB = 8;          // entries per batch: 16 row loads per lane in flight
  for (int e0 = 0; e0 < nlist; e0 += EB) {
    int key[EB];
    float4 av[EB], bv[EB];
#pragma unroll
    for (int q = 0; q < EB; ++q) {
      if (e0 + q >= nlist) break;                 // wave-uniform
      key[q] = __builtin_amdgcn_readfirstlane(s_list[wv][e0 + q]);
      const int rp = (key[q] >> 16) * 32 + ((key[q] >> 5) & 31), col = (u0 + ((key[q] >> 10) & 63)) * 32 + (key[q] & 31);
      const long ro = ((long)b * a.L + panel * kPanelRows + rp) * a.c_in, co = ((long)b * a.S + col) * a.c_in;
      const bool in = lane * 4 < a.c_in;
      av[q] = bv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in && a.in_dtype == FM_F32) {
        av[q] = reinterpret_cast<const float4*>((const float*)a.src0 + ro)[lane];
        bv[q] = reinterpret_cast<const float4*>((const float*)a.src1 + co)[lane];
      } else if (in) {       // float16 / bfloat16 rows: exact in float32, products of two halves are exact too
        av[q] = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)a.src0 + ro)[lane], a.in_dtype);
        bv[q] = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)a.src1 + co)[lane], a.in_dtype);
      }
    }
    float x[EB];
#pragma unroll
    for (int q = 0; q < EB; ++q) {
      if (e0 + q >= nlist) break;
      float p = av[q].x * bv[q].x;
      p = __builtin_fmaf(av[q].y, bv[q].y, p);
      p = __builtin_fmaf(av[q].z, bv[q].z, p);
      p = __builtin_fmaf(av[q].w, bv[q].w, p);
      x[q] = wave_reduce64<true>(p);          // the exact float32 dot product, same bits in every lane
    }
#pragma unroll
    for (int q = 0; q < EB; ++q) {
      if (e0 + q >= nlist) break;
      const int kk2 = key[q] >> 16, ul = (key[q] >> 10) & 63, rl = (key[q] >> 5) & 31, cl = key[q] & 31;
      const float nr = s_nmr[kk2][rl];
      const float nc = s_nmc[ul * 32 + cl];
      const float rr = __builtin_fmaf(x[q], a.k, nr);
      const float cc = __builtin_fmaf(x[q], a.k, nc);
      if (lane == 0) {
        s_rowacc[wv * kPanelRows + kk2 * 32 + rl] += __builtin_amdgcn_exp2f(rr);
        s_colacc[wv * cpitch + ul * 32 + cl] += __builtin_amdgcn_exp2f(cc);
      }
      if (rr > a.lt && cc > a.lt) {                     // wave-uniform: a candidate (superset of conf > thr)
        const int col = (u0 + ul) * 32 + cl;
        if (qn < kSparseQueue) {
          if (lane == 0) { s_qkey[wv][qn] = kk2 * 32 + rl; s_qcol[wv][qn] = col; s_qx[wv][qn] = x[q]; }
        } else if (lane == 0) {                         // queue full: straight to the slot lists
          record(kk2 * 32 + rl, col, x[q]);
        }
        ++qn;
      }
    }
  }

  DIAG_STAMP(5)
  // ---- results of this wave: candidates, dense flags ----
  {
    const int nq = min(qn, kSparseQueue);
    if (lane < nq) record(s_qkey[wv][lane], s_qcol[wv][lane], s_qx[wv][lane]);
  }
  // ---- the workgroup's dense-unit count (one atomic per workgroup) and its row / column partials: the 8 waves'
  // accumulators folded in a fixed order ----
  if (lane == 0) s_hot[wv] = nd_units;
  __syncthreads();
  if (tid == 0) {
    int nd = 0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) nd += s_hot[w8];
    if (nd) {
      atomicAdd(&a.dense_cnt[b], nd);
      atomicAdd(&a.scal->dense_units, nd);
      if (!a.dense_enabled) atomicOr(&a.scal->flags, (unsigned)FM_DEV_DENSE);   // nobody will redo this sample
    }
  }
  if (tid < kPanelRows) {
    float s2 = s_rowacc[tid];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) s2 += s_rowacc[w8 * kPanelRows + tid];
    a.rowS[((long)b * a.splits + split) * a.Lp + panel * kPanelRows + tid] = s2;
  }
  float* co = a.colS + ((long)b * a.panels + panel) * a.Sp + u0 * 32;
  for (int c = tid; c < U * 32; c += 512) {
    float s2 = s_colacc[c];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) s2 += s_colacc[w8 * cpitch + c];
    co[c] = s2;
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

hipError_t launch_sum_sparse(const void* feat0, const void* feat1, int in_dtype, int c_in, const CoarseWs& w, char* base,
                             float inv_ct, float thr, int dense_enabled, int allow_dead, hipStream_t st) {
  SparseArgs a;
  a.in_dtype = in_dtype;
  a.q0 = (const signed char*)(base + w.q0); a.q1 = (const signed char*)(base + w.q1);
  a.src0 = feat0; a.src1 = feat1; a.c_in = c_in;
  a.rowmax_u = (const unsigned*)(base + w.rowmax_u); a.colmax_u = (const unsigned*)(base + w.colmax_u);
  a.sigimg = (const float*)(base + w.sigimg);
  a.l1_0 = (const float*)(base + w.l1_0); a.l1_1 = (const float*)(base + w.l1_1);
  a.bstat0 = (const float4*)(base + w.bstat0); a.bstat1 = (const float4*)(base + w.bstat1);
  a.umax = (const float*)(base + w.umax);
  a.nmr = (float*)(base + w.nmr); a.nmc = (float*)(base + w.nmc); a.emarg = (float*)(base + w.emarg);
  a.rowS = (float*)(base + w.rowS); a.colS = (float*)(base + w.colS);
  a.dense_cnt = (int*)(base + w.dense_cnt); a.scal = (Scalars*)(base + w.scalars);
  a.diag = (float*)(base + w.rowB);      // (diagnostic builds run on a full-size workspace)
  a.cand_count = (int*)(base + w.cand_count); a.cand_j = (int*)(base + w.cand_j); a.cand_x = (float*)(base + w.cand_x);
  a.ccand_count = (int*)(base + w.ccand_count); a.ccand_i = (int*)(base + w.ccand_i); a.ccand_x = (float*)(base + w.ccand_x);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.splits = w.splits_s; a.units_s = w.units_s;
  a.slots = w.slots; a.dense_enabled = dense_enabled; a.allow_dead = allow_dead;
  {
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.units_s * 32) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
  }
  a.k = inv_ct * kLog2e; a.lt = log2f(thr); a.inv_ct = inv_ct; a.cpad = (float)w.C;
  const int blocks = w.N * a.splits * w.panels;
  const int smem = a.units_s * 32 * 10 * 4;      // column stabilisers + steps + 8 waves' column accumulators
  hipError_t e = hipSuccess;
#define FM_SPARSE_CASE(CC)                                                                   \
  case CC: {                                                                                 \
    static unsigned long long lds_set = 0;                                                   \
    e = ensure_dynamic_lds(&k_sum_sparse<CC>, kUnitsPerSplit * 32 * 10 * 4, &lds_set);        \
    if (e != hipSuccess) return e;                                                           \
    hipLaunchKernelGGL(k_sum_sparse<CC>, dim3(blocks), dim3(512), smem, st, a);              \
    break;                                                                                   \
  }
  switch (w.C) {
    FM_SPARSE_CASE(64)
    FM_SPARSE_CASE(128)
    FM_SPARSE_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_SPARSE_CASE
  return hipGetLastError();
}

}  // namespace fm
