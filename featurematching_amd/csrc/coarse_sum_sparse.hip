// Coarse stage, the SPARSE sum kernel: dual-softmax denominators and the candidate list from the units
// that matter, without barriers inside the sweep.
//
// Reproduces network/utils/coarse_matching_new.py:64-68 (correlation + dual softmax) for every entry whose
// term is not negligible.  After the max pass (k_corr<C,0>: row / column maxima and the maximum of every
// 32 x 32 unit of the float16 hi x hi product) each entry s_ij falls in one of three classes:
//   * negligible : k x~_ij + margin lies more than 2^32 below BOTH its row's and its column's stabiliser: it adds
//                  < 2^-32 to sums that are >= e^-2E ~ 1 (<= S 2^-32 ~ 1e-6 relative in total, inside the 1e-5
//                  parity bar) and cannot be a candidate.  Never touched again.  With dual-softmax-trained
//                  (peaked) descriptors that is all but ~1 entry per row.
//   * significant, few per unit : the exact float32 dot product of the two descriptors is recomputed from the
//                  caller's rows (one wave, 4 channels per lane, fixed butterfly) and enters
//                  sum_j exp2(k x - m^_i), sum_i exp2(k x - c^_j) and, when both terms pass log2 thr, the
//                  candidate list - the same number in numerator and denominator of conf, as in the reference.
//   * significant, many per unit (flat similarity: untrained network, repetitive texture) : the unit is FLAGGED
//                  for the dense sum kernel (k_corr<C,1>: float32-equivalent hi/lo product on the matrix cores for
//                  all 1024 entries), which runs next and exits at once when nothing is flagged.
//
// Structure: a workgroup = 8 INDEPENDENT waves = 8 row blocks (32 rows) x one range of <= 16 column units; no
// LDS tile ring and no barrier in the sweep - with ~1 unit in 5 alive, lock-stepping 8 waves through shared
// tiles cost the old sum sweep 19k of its 55k cycles in barrier waits at 640x480.  A wave keeps its 32 rows as
// hi A-fragments in 64 VGPRs (no lo plane: half the prologue bytes), decides from the unit maxima which of its
// units are alive, and for each of them loads the hi B-fragments straight into registers (16 x 1 KiB contiguous
// blocks of the fragment-major plane, the next unit's issued behind the MFMA chain), runs 16 MFMAs, screens the
// 32 x 32 accumulators against the stabilisers and resolves the significant entries exactly.  All sums are
// formed in a fixed order: results are bitwise reproducible.
#include "fm_device.h"

namespace fm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxExact = 24;         // significant entries of a unit resolved by exact dot products; more -> dense kernel
constexpr int kSparseQueue = 64;      // candidates a wave parks in LDS (one per lane at the hand-over)

struct SparseArgs {
  const _Float16* hi0; const _Float16* hi1;
  const float* src0; const float* src1; int c_in;      // the caller's descriptors [N,L,c_in] / [N,S,c_in]
  const unsigned* rowmax_u; const unsigned* colmax_u;  // max pass: ord_encode'd maxima of the f16 product
  const float* norm0; const float* norm1; const float* bmax0; const float* bmax1;
  const float* umax;
  float* nmr; float* nmc; float* emarg;                // written here: stabilisers, pair margin
  float* rowS; float* colS;                            // partial sums [N][splits][Lp], [N][panels][Sp]
  float* dense_map; Scalars* scal;
  float* diag;                                         // diagnostic build: stamp buffer
  int* cand_count; int* cand_j; float* cand_x;
  int L, S, Lp, Sp, panels, splits, units_s, slots, pgroup;
  float k, lt, inv_ct, sqrt_c;
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

template <int C>
__global__ __launch_bounds__(512) void k_sum_sparse(SparseArgs a) {
  constexpr int KSTEPS = C / 16;
  constexpr int LIST = kUnitsPerSplit * kMaxExact;      // significant entries a wave can park
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
  const int U = max(0, min(a.units_s, nunits - u0));          // units of this workgroup's range (<= 16)
  const int rb = panel * 8 + wv;                              // this wave's row block
  const int wrow0 = rb * 32;
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps per phase of every wave (tools/diag_sparse.py)
  unsigned long long dg[8];
  dg[0] = __builtin_amdgcn_s_memtime();
#define DIAG_STAMP(i) dg[i] = __builtin_amdgcn_s_memtime();
#else
#define DIAG_STAMP(i)
#endif

  __shared__ float s_nmr[8][32];
  __shared__ float s_nmc[kUnitsPerSplit * 32];
  __shared__ float s_cmax[kUnitsPerSplit];
  __shared__ int s_hot[8];
  __shared__ float s_colacc[8][kUnitsPerSplit * 32];
  __shared__ int s_list[8][LIST];              // (unit << 10) | (row in wave << 5) | column in unit
  __shared__ int s_qkey[8][kSparseQueue];
  __shared__ float s_qx[8][kSparseQueue];

  // ---- everything the decisions below depend on is requested at once (one memory round trip): the block norm
  // maxima, this wave's row statistics, the range's column statistics, the unit maxima and, speculatively, the
  // wave's A fragments (lane (r,h): row r, k = h*C/2 + 8*ks + 0..7; one contiguous 1 KiB block per k-step of the
  // fragment-major plane of k_prep_split) ----
  float own = 0.f, oth = 0.f;          // largest descriptor norm of image 0 / image 1 (every wave folds them itself)
  for (int i = lane; i < a.Lp / 32; i += 64) own = fmaxf(own, a.bmax0[(long)b * (a.Lp / 32) + i]);
  for (int i = lane; i < nunits; i += 64) oth = fmaxf(oth, a.bmax1[(long)b * nunits + i]);
  const long gi = (long)b * a.Lp + wrow0 + r;
  const unsigned rmax_u = a.rowmax_u[gi];
  const float rnorm = a.norm0[gi];
  unsigned cmax_u = 0u;
  float cnorm = 0.f;
  const long gj = (long)b * a.Sp + u0 * 32 + tid;
  if (tid < U * 32) { cmax_u = a.colmax_u[gj]; cnorm = a.norm1[gj]; }
  float um = -INFINITY;
  if (lane < U) um = a.umax[((long)b * (a.Lp / 32) + rb) * nunits + u0 + lane];
  half8 ahi[KSTEPS];
  if (wrow0 < a.L) {
    const long off = (((long)b * a.Lp + wrow0) / 32 * KSTEPS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) ahi[ks] = *reinterpret_cast<const half8*>(a.hi0 + off + ks * 512);
  } else {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) ahi[ks][e] = (_Float16)0.f;
  }
  for (int c = lane; c < U * 32; c += 64) s_colacc[wv][c] = 0.f;

  own = wave_reduce64<false>(own);
  oth = wave_reduce64<false>(oth);
  DIAG_STAMP(1)
  const float emarg = pair_margin_log2(own, oth, a.inv_ct, a.sqrt_c);
  if (panel == 0 && split == 0 && tid == 0) {
    a.emarg[b] = emarg;
    // a prep workgroup that met NaN/Inf/|x| >= 32768 reported +inf
    if (!(own < INFINITY) || !(oth < INFINITY)) atomicOr(&a.scal->flags, (unsigned)FM_DEV_RANGE);
  }

  // ---- stabilisers: this wave's 32 rows, the workgroup's column range ----
  // lanes 0..31 (and their mirror 32..63): -stabiliser*log2e of row wrow0 + r
  const float nm_lane = neg_stabiliser_log2(ord_decode(rmax_u), rnorm, oth, a.inv_ct, a.sqrt_c);
  if (h == 0) {
    s_nmr[wv][r] = nm_lane;
    if (split == 0) a.nmr[gi] = nm_lane;
  }
  // padded rows (>= L) never contribute: keep them out of the wave's largest stabiliser
  const float wmax_nmr = wave_reduce64<false>(wrow0 + r < a.L ? nm_lane : -INFINITY);
  if (tid < U * 32) {
    const float nm = neg_stabiliser_log2(ord_decode(cmax_u), cnorm, own, a.inv_ct, a.sqrt_c);
    s_nmc[tid] = nm;
    if (panel == 0) a.nmc[gj] = nm;
  }
  __syncthreads();
  for (int u = wv; u < U; u += 8) {
    const float v = wave_reduce64<false>((u0 + u) * 32 + r < a.S ? s_nmc[u * 32 + r] : -INFINITY);
    if (lane == 0) s_cmax[u] = v;
  }
  __syncthreads();

  // ---- which of this wave's units are alive (same bound as the dense kernel's block-sparse skip) ----
  bool hot = false;
  if (lane < U && wrow0 < a.L) {
    const float top = __builtin_fmaf(um, a.k, emarg);             // >= k * (exact product), log2 domain
    hot = !((top + wmax_nmr < -kSkipLog2) && (top + s_cmax[lane] < -kSkipLog2));
  }
  unsigned mask = (unsigned)__ballot(hot);          // wave-uniform
  if (lane == 0) s_hot[wv] = __builtin_popcount(mask);
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) tot += s_hot[w8];
  unsigned dmask = 0;            // units left to the dense kernel
  if (tot * 2 > 8 * U) {         // more than half of the block is alive: flat similarity, a matrix-core job
    dmask = mask;
    mask = 0;
  }

  int nlist = 0;                 // parked significant entries (wave-uniform)
  DIAG_STAMP(2)
#ifdef FM_DIAG_CLOCK
  const int diag_units = __builtin_popcount(mask);
  dg[3] = dg[2];
#endif
  if (mask) {
    // this lane's 16 row stabilisers (rows 8q + 4h + 0..3 of the wave's 32) with the pair margin folded in
    float nmr_e[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) nmr_e[g] = s_nmr[wv][(g & 3) + 8 * (g >> 2) + 4 * h] + emarg;
    const bool row_edge = (wrow0 + 32 > a.L);

    half8 bfr[KSTEPS];
    auto load_b = [&](int ul) {
      const _Float16* src = a.hi1 + (((long)b * nunits + u0 + ul) * KSTEPS * 64 + lane) * 8;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) bfr[ks] = *reinterpret_cast<const half8*>(src + ks * 512);
    };
    load_b(__builtin_ctz(mask));
#ifdef FM_DIAG_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DIAG_STAMP(3)
#endif
    while (mask) {
      const int ul = __builtin_ctz(mask);
      mask &= mask - 1;
      const int ucol0 = (u0 + ul) * 32;
      // The accumulator starts at 0, or at -inf for padded rows (>= L) / padded columns (>= S): such entries
      // stay -inf through the chain and fail the significance test.
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0.f;
      if (row_edge || ucol0 + 32 > a.S) {
        const float cb = (ucol0 + r < a.S) ? 0.f : -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = (wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h < a.L) ? cb : -INFINITY;
      }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bfr[ks], acc, 0, 0, 0);
      if (mask) load_b(__builtin_ctz(mask));        // next unit's fragments fly behind the epilogue

      // significance: k x~ + margin within 2^32 of the row's or the column's stabiliser
      const float nmc_e = s_nmc[ul * 32 + r] + emarg;
      unsigned lm = 0;           // bit g: accumulator register g of this lane is significant
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float v = __builtin_fmaf(acc[g], a.k, fmaxf(nmr_e[g], nmc_e));
        lm |= (v > -kSkipLog2) ? (1u << g) : 0u;
      }
      unsigned long long any = __ballot(lm != 0);
      if (!any) continue;
      const int nsig = (int)wave_reduce64<true>((float)__builtin_popcount(lm));      // <= 1024: exact in float
      if (nsig > kMaxExact) { dmask |= 1u << ul; continue; }
      // park the entries (fixed order: lane, then register); they are resolved after the sweep, several at a time
      while (any) {
        const int l = __builtin_ctzll(any);
        any &= any - 1;
        unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)lm, l);
        while (bits) {
          const int g = __builtin_ctz(bits);
          bits &= bits - 1;
          const int rl = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
          if (lane == 0) s_list[wv][nlist] = (ul << 10) | (rl << 5) | (l & 31);
          ++nlist;
        }
      }
    }
  }

  // ---- the parked entries: exact float32 dot products of the caller's descriptors, four entries (eight row
  // loads per lane) in flight; sums in list order (deterministic) ----
  float racc = 0.f;              // lanes 0..31: sum_j exp2(k x - m^) of row wrow0 + lane over this range
  int qn = 0;                    // parked candidates (wave-uniform)
  DIAG_STAMP(4)
  for (int e0 = 0; e0 < nlist; e0 += 4) {
    int key[4];
    float4 av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      key[q] = __builtin_amdgcn_readfirstlane(s_list[wv][min(e0 + q, nlist - 1)]);
      const int rl = (key[q] >> 5) & 31, col = (u0 + (key[q] >> 10)) * 32 + (key[q] & 31);
      const float4* rp = reinterpret_cast<const float4*>(a.src0 + ((long)b * a.L + wrow0 + rl) * a.c_in);
      const float4* cp = reinterpret_cast<const float4*>(a.src1 + ((long)b * a.S + col) * a.c_in);
      const bool in = lane * 4 < a.c_in;
      av[q] = in ? rp[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[q] = in ? cp[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float p = av[q].x * bv[q].x;
      p = __builtin_fmaf(av[q].y, bv[q].y, p);
      p = __builtin_fmaf(av[q].z, bv[q].z, p);
      p = __builtin_fmaf(av[q].w, bv[q].w, p);
      x[q] = wave_reduce64<true>(p);          // the exact float32 dot product, same bits in every lane
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (e0 + q >= nlist) break;
      const int ul = key[q] >> 10, rl = (key[q] >> 5) & 31, cl = key[q] & 31;
      const float nr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nm_lane), rl));
      const float nc = s_nmc[ul * 32 + cl];
      const float rr = __builtin_fmaf(x[q], a.k, nr);
      const float cc = __builtin_fmaf(x[q], a.k, nc);
      racc += (lane == rl) ? __builtin_amdgcn_exp2f(rr) : 0.f;
      if (lane == 0) s_colacc[wv][ul * 32 + cl] += __builtin_amdgcn_exp2f(cc);
      if (rr > a.lt && cc > a.lt) {                     // wave-uniform: a candidate (superset of conf > thr)
        const int col = (u0 + ul) * 32 + cl;
        if (qn < kSparseQueue) {
          if (lane == 0) { s_qkey[wv][qn] = (col << 5) | rl; s_qx[wv][qn] = x[q]; }
        } else if (lane == 0) {                         // queue full: straight to the row's slot list
          const long grow = (long)b * a.Lp + wrow0 + rl;
          const int pos = atomicAdd(&a.cand_count[grow], 1);
          if (pos < a.slots) { a.cand_j[grow * a.slots + pos] = col; a.cand_x[grow * a.slots + pos] = x[q]; }
          else atomicOr(&a.scal->flags, (unsigned)FM_INT_SCREEN_OVERFLOW);
        }
        ++qn;
      }
    }
  }

  DIAG_STAMP(5)
  // ---- results of this wave: row partial, dense flags, candidates ----
  if (lane < 32) a.rowS[((long)b * a.splits + split) * a.Lp + wrow0 + lane] = racc;
  if (lane < U) a.dense_map[((long)b * (a.Lp / 32) + rb) * nunits + u0 + lane] = ((dmask >> lane) & 1u) ? 1.f : 0.f;
  {
    const int nq = min(qn, kSparseQueue);
    if (lane < nq) {
      const int key = s_qkey[wv][lane];
      const long grow = (long)b * a.Lp + wrow0 + (key & 31);
      const int pos = atomicAdd(&a.cand_count[grow], 1);
      if (pos < a.slots) { a.cand_j[grow * a.slots + pos] = key >> 5; a.cand_x[grow * a.slots + pos] = s_qx[wv][lane]; }
      else atomicOr(&a.scal->flags, (unsigned)FM_INT_SCREEN_OVERFLOW);
    }
  }
  // ---- the workgroup's dense-unit count (one atomic per workgroup) and its column partial: the 8 waves'
  // accumulators folded in a fixed order ----
  if (lane == 0) s_hot[wv] = __builtin_popcount(dmask);
  __syncthreads();
  if (tid == 0) {
    int nd = 0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) nd += s_hot[w8];
    if (nd) atomicAdd(&a.scal->dense_units, nd);
  }
  float* co = a.colS + ((long)b * a.panels + panel) * a.Sp + u0 * 32;
  if (tid < U * 32) {
    float s = s_colacc[0][tid];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) s += s_colacc[w8][tid];
    co[tid] = s;
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

hipError_t launch_sum_sparse(const float* feat0, const float* feat1, int c_in, const CoarseWs& w, char* base,
                             float inv_ct, float thr, hipStream_t st) {
  SparseArgs a;
  a.hi0 = (const _Float16*)(base + w.hi0); a.hi1 = (const _Float16*)(base + w.hi1);
  a.src0 = feat0; a.src1 = feat1; a.c_in = c_in;
  a.rowmax_u = (const unsigned*)(base + w.rowmax_u); a.colmax_u = (const unsigned*)(base + w.colmax_u);
  a.norm0 = (const float*)(base + w.norm0); a.norm1 = (const float*)(base + w.norm1);
  a.bmax0 = (const float*)(base + w.bmax0); a.bmax1 = (const float*)(base + w.bmax1);
  a.umax = (const float*)(base + w.umax);
  a.nmr = (float*)(base + w.nmr); a.nmc = (float*)(base + w.nmc); a.emarg = (float*)(base + w.emarg);
  a.rowS = (float*)(base + w.rowS); a.colS = (float*)(base + w.colS);
  a.dense_map = (float*)(base + w.dense_map); a.scal = (Scalars*)(base + w.scalars);
  a.diag = (float*)(base + w.rowB);
  a.cand_count = (int*)(base + w.cand_count); a.cand_j = (int*)(base + w.cand_j); a.cand_x = (float*)(base + w.cand_conf);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.splits = w.splits_s; a.units_s = w.units_s;
  a.slots = w.slots;
  {
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.units_s * 32) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
  }
  a.k = inv_ct * kLog2e; a.lt = log2f(thr); a.inv_ct = inv_ct; a.sqrt_c = sqrtf((float)w.C);
  const int blocks = w.N * a.splits * w.panels;
  switch (w.C) {
    case 64: hipLaunchKernelGGL(k_sum_sparse<64>, dim3(blocks), dim3(512), 0, st, a); break;
    case 128: hipLaunchKernelGGL(k_sum_sparse<128>, dim3(blocks), dim3(512), 0, st, a); break;
    case 256: hipLaunchKernelGGL(k_sum_sparse<256>, dim3(blocks), dim3(512), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace fm
