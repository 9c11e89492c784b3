// Coarse stage, the tile-based correlation sweep of rounds 1-3.  What still runs from here is pass C (MODE 2), the exact
// re-screening of FM_MODE_EXACT_SCREENING; passes A, B and D have been replaced by k_max_i8 (coarse_max_i8.hip), k_screen /
// k_dense (coarse_screen.hip, coarse_dense.hip) and k_dense<C, CONF>; their description is kept because pass C shares
// their structure.
//
// Reproduces network/utils/coarse_matching_new.py:64-68 (all-pairs correlation + dual
// softmax statistics) without ever writing the L x S matrix:
//
//   pass A (MODE 0): raw = hi0 . hi1^T on the f16 matrix cores; per-row and per-column
//                    maxima (partial over column splits / row panels) and the maximum of
//                    every 32 x 32 unit (block map for pass B).
//   pass B (MODE 1): raw = hi0.hi1 + lo0.hi1 + hi0.lo1 (float32-accurate product);
//                    row sums  sum_j exp(s_ij - m^_i), column sums sum_i exp(s_ij - c^_j)
//                    with the stabilisers of k_reduce<0>, and the sparse candidate list
//                    {(i,j): s_ij - m^_i > ln thr  and  s_ij - c^_j > ln thr}, a superset
//                    of every entry with conf > thr (coarse_matching_new.py:99).
//                    BLOCK-SPARSE: a unit whose largest entry (from pass A, plus the f16 error
//                    margin) lies more than 2^32 below every row stabiliser of its 32 rows AND every
//                    column stabiliser of its 32 columns is skipped: each of its entries adds
//                    < 2^-32 to a sum that is >= e^-2E ~ 1 (at most S * 2^-32 ~ 1e-6 relative in total,
//                    inside the 1e-5 parity bar) and none can be a candidate.  With dual-softmax-
//                    trained (peaked) descriptors ~81 % of the units qualify; with flat similarity
//                    nothing is skipped and the sweep is dense.
//   pass C (MODE 2): only when pass B's screening overflowed a row's candidate slots (flat
//                    similarity rows, e.g. an untrained network): the same product again, screened
//                    with the now-known softmax denominators, log2 P_row > log2 thr and
//                    log2 P_col > log2 thr, which at most 1/thr entries of a row can pass.
//                    The kernel is always enqueued and exits at once when it is not needed.
//   pass D (MODE 3): on request only: the dense conf_matrix [N,L,S] (coarse_matching_new.py:70), written
//                    from the same product and the log-softmax offsets; the training loss consumes it.
//
// Structure (one workgroup = 8 waves = 256 rows of image 0; cf. SURVEY.md 7, hard part 2):
//   * each wave keeps its 32 rows x C of image-0 descriptors as MFMA A-fragments in
//     registers for the whole sweep (no K loop, no re-read);
//   * image-1 descriptors stream through LDS in 64-column tiles - a 4-deep ring of hi tiles in the max pass,
//     a double buffer of hi+lo tiles otherwise - filled by LDS-DMA (global_load_lds_dwordx4) one 1 KiB
//     fragment block at a time, handed over by counted vmcnt + raw s_barrier; planes are fragment-major
//     (k_prep_split), so the LDS image is lane-linear and every ds_read_b128 is bank-conflict free; the
//     tile's metadata (column stabilisers, unit maxima) arrives by 4-byte LDS-DMA next to it;
//   * inside the sweep every LDS access is inline asm and no register carries a global load across
//     iterations: hipcc would otherwise order them against the ring's DMAs with vmcnt(0) (see below);
//   * the f32 accumulator tile (32 rows x 32 cols per wave and unit) never leaves registers:
//     the epilogue turns it into exp2 terms, accumulates row sums in registers across
//     the whole sweep and reduces column sums lane-locally (rows live in registers,
//     columns on lanes: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5));
//     the 8 waves' column partials of a tile meet in LDS and one wave folds them (fixed order) into the
//     workgroup's single column partial per panel.
#include <stdlib.h>

#include "fm_device.h"

namespace fm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// B-fragment prefetch distance (k-steps) in the sum / max sweeps
#ifndef FM_PF_SUM
#define FM_PF_SUM 2
#endif
#ifndef FM_PF_MAX
#define FM_PF_MAX 4
#endif
#ifndef FM_NBUF_MAX
#define FM_NBUF_MAX 4       // LDS tile ring depth of the max pass (32 KiB hi tiles at C = 256)
#endif
constexpr int kCandQueue = 64;        // candidates a wave parks in LDS per sweep (one per lane at the hand-over)

struct CorrArgs {
  const _Float16* hi0; const _Float16* lo0; const _Float16* hi1; const _Float16* lo1;
  const float* nmr; const float* nmc;
  float* rowpart; float* colpart;
  unsigned* rowmax_u; unsigned* colmax_u;   // MODE 0: row / column maxima of the f16 product (ord_encode, atomicMax)
  float* umax;            // [N][Lp/32][Sp/32] unit maxima of the raw f16 product (written by MODE 0)
  const int* dense_cnt;   // [N] units per sample the sparse sum kernel left to the dense one (> 0: MODE 1 redoes the sample)
  const int* dense_units; // their total (0: MODE 1 has nothing to do)
  const float* emarg;     // [N] log2-domain bound of |screening product - exact product| * k
  const float* f16inv;    // [N] 1 / (power-of-two scales of the two images' float16 planes): accumulator -> dot product
  const float* sigimg;    // [N][2] int8 steps of the two images: unit maxima are integer screening products
  int* cand_count; int* cand_j; float* cand_x; unsigned* flags;
  int* ccand_count; int* ccand_i; float* ccand_x;      // the same candidates listed per column (rows, dot products)
  int* cand_count_b; int* cand_j_b; float* cand_x_b;   // the dense kernel's candidate set (samples it redid)
  int* ccand_count_b; int* ccand_i_b; float* ccand_x_b;
  float* conf;            // MODE 3: dense [N,L,S] output
  int L, S, Lp, Sp, panels, tiles, splits, tiles_per_split, slots;
  int pgroup;             // panels per XCD-locality group of the workgroup order
  float k;    // log2(e) / (C*T): raw dot product -> log2-domain similarity
  float lt;   // log2(thr)
};

// Workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of
// logical ids so that the workgroups sharing a column stream share an L2 (speed only).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds4(const float* gsrc, float* lds_wave_base) {      // lane l -> base + 4*l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xf, BANK, false));
}
// max over the 64 lanes, in every lane; DPP + permlane swaps (no LDS).  See fine.hip for the lane maps.
__device__ __forceinline__ float wave_max64(float v) {
  v = fmaxf(v, dpp_mov<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_mov<0x4E, 0xf>(v, v));
  { float t = dpp_mov<0x104, 0x5>(v, v); t = dpp_mov<0x114, 0xA>(t, v); v = fmaxf(v, t); }
  v = fmaxf(v, dpp_mov<0x128, 0xf>(v, v));
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = fmaxf(p, q); }
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = fmaxf(p, q); }
  return v;
}

// both halves of the wave combine their values (lane l with lane l^32); v_permlane32_swap, no LDS crossbar
template <bool SUM>
__device__ __forceinline__ float combine_halves(float v) {
  float p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return SUM ? p + q : fmaxf(p, q);
}

// reduction over the 32 lanes that share lane>>5 (max or sum), result in every lane of the half; DPP only
template <bool SUM>
__device__ __forceinline__ float half_reduce32(float v) {
  auto op = [](float x, float y) { return SUM ? x + y : fmaxf(x, y); };
  v = op(v, dpp_mov<0xB1, 0xf>(v, v));                                                       // lane ^ 1
  v = op(v, dpp_mov<0x4E, 0xf>(v, v));                                                       // lane ^ 2
  { float t = dpp_mov<0x104, 0x5>(v, v); t = dpp_mov<0x114, 0xA>(t, v); v = op(v, t); }     // lane ^ 4
  v = op(v, dpp_mov<0x128, 0xf>(v, v));                                                      // lane ^ 8
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = op(p, q); }   // ^ 16
  return v;
}

// LDS accesses of the small tables INSIDE the sweep are inline asm: hipcc orders every LDS access it can
// see against the outstanding LDS-DMA writes of the tile ring (it cannot tell the objects apart) and put
// `s_waitcnt vmcnt(0)` in front of each one - i.e. waited for the tile prefetch it had just issued, which
// serialised the ring (~1 us per tile).  The waits below are ours; outputs are tied to them ("+v").
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ unsigned lds_addr(T* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) T*)p;
}
__device__ __forceinline__ void lds_store_b32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store_b32(unsigned addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int C, int MODE>
__global__ __launch_bounds__(512) void k_corr(CorrArgs a) {
  constexpr int KSTEPS = C / 16;
  constexpr int ROWB = C * 2;
  constexpr int PLANES = MODE ? 2 : 1;
  constexpr int PLANE_BYTES = kTileCols * ROWB;
  constexpr int BUF_BYTES = PLANES * PLANE_BYTES;
  // LDS tile ring: the max pass needs only the hi plane (32 KiB per tile at C = 256), so it keeps 3 tiles
  // in flight; the other passes (hi + lo) double-buffer.
  constexpr int NBUF = MODE ? 2 : FM_NBUF_MAX;
  constexpr int GLDS_PER_TILE = PLANES * (PLANE_BYTES / 1024 / 8);     // LDS-DMA instructions per wave and tile
  constexpr int INSTR_PER_WAVE = PLANE_BYTES / 1024 / 8;
  constexpr bool SPARSE = (MODE == 1 || MODE == 2);
  constexpr int META = 80;          // per tile: 64 column stabilisers + 16 unit maxima (8 waves x 2 units)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  if (MODE == 2 && !(*a.flags & FM_INT_SCREEN_OVERFLOW)) return;   // uniform: fast screening sufficed
  if (MODE == 1 && *a.dense_units == 0) return;                    // uniform: the sparse sum kernel handled every unit
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps (s_memtime) per phase of every wave
  const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long diag_mfma = 0, diag_epi = 0, diag_bar = 0, diag_pro = 0, diag_loop_end = 0;
  int diag_units = 0, diag_cand = 0;
#define DIAG_T0 const unsigned long long diag_t = __builtin_amdgcn_s_memtime();
#define DIAG_ADD(x) x += __builtin_amdgcn_s_memtime() - diag_t;
#else
#define DIAG_T0
#define DIAG_ADD(x)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // Logical workgroup order: sample, then GROUPS of a.pgroup panels, inside a group split-major.  An XCD's
  // contiguous share of that order (xcd_remap) is then a compact block of a few panels x a few splits:
  // its L2 has to pull only those panels' A fragments and those splits' column ranges over the fabric
  // (panel-fastest order made every XCD fetch ALL of image 0 during the prologue: 14.5k cycles at 640x480).
  int kk = xcd_remap(blockIdx.x, gridDim.x);
  const int per_sample = a.panels * a.splits;
  const int b = kk / per_sample;
  kk -= b * per_sample;
  const int gsz = a.pgroup * a.splits;
  const int pg = kk / gsz;
  kk -= pg * gsz;
  const int pcount = min(a.pgroup, a.panels - pg * a.pgroup);
  const int split = kk / pcount;
  const int panel = pg * a.pgroup + (kk - split * pcount);
  const int t0 = split * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.tiles);

  if (MODE == 1 && a.dense_cnt[b] == 0) return;      // uniform: this sample was handled by the screening kernel
  // (its lists hold every significant entry and never overflow - a row or column with more than `slots` of them sends
  // the sample here in MODE 1 - so the exact screening only ever refills the lists of samples the dense kernel redid)
  if (MODE == 2 && a.dense_cnt[b] == 0) return;
  const float inv_sc = a.f16inv[b];                  // the planes carry exact power-of-two scales (k_prep_f16)
  const float kq = a.k * inv_sc;                     // accumulator -> log2-domain similarity
  // candidate set of this sample: the dense kernel's own for the samples it redoes
  const bool dense_sample = MODE == 1 || (MODE == 2 && a.dense_cnt[b] != 0);
  int* const cand_count = dense_sample ? a.cand_count_b : a.cand_count;
  int* const cand_j = dense_sample ? a.cand_j_b : a.cand_j;
  float* const cand_x = dense_sample ? a.cand_x_b : a.cand_x;
  int* const ccand_count = dense_sample ? a.ccand_count_b : a.ccand_count;
  int* const ccand_i = dense_sample ? a.ccand_i_b : a.ccand_i;
  float* const ccand_x = dense_sample ? a.ccand_x_b : a.ccand_x;
  // unit maxima of the max pass are integer screening products: x~ = sigma_0 sigma_1 (q_i . q_j)
  const float kss = SPARSE ? a.k * a.sigimg[b * 2] * a.sigimg[b * 2 + 1] : 0.f;
  const _Float16* planes1[2] = {a.hi1 + (long)b * a.Sp * C, a.lo1 + (long)b * a.Sp * C};   // fragment-major
  // Small per-workgroup tables live in their OWN static LDS objects, not in the dynamic tile ring: hipcc
  // orders every LDS access it can see against outstanding LDS-DMA writes it cannot tell apart from it -
  // with the tables inside `smem` it put `s_waitcnt vmcnt(0)` in front of each table read, i.e. waited for
  // the tile prefetch it had just issued (the whole ring was serialised: ~1 us per tile).
  __shared__ float s_nmr[8 * 32];                 // row stabilisers of the 8 waves' rows
  __shared__ float s_meta[2 * META];              // per tile: 64 column stabilisers + 16 unit maxima
  __shared__ float s_colred[2 * 8 * 64];          // per tile parity: the 8 waves' column partials of 64 columns
  // wave-private candidate queue (sum / screening passes): candidates found during the sweep are parked
  // here and handed to the global per-row slot lists once, after the sweep.  (A global atomicAdd with
  // return per find stalls the wave for a memory round trip in the middle of the MFMA pipeline: 10 us of
  // a 43 us sweep at 640x480, where nearly every computed unit holds a match.)
  __shared__ int s_qkey[8 * kCandQueue];          // (col << 5) | local row
  __shared__ float s_qx[8 * kCandQueue];
  __shared__ int s_qcnt[8];
  float* nmr_lds = s_nmr + wv * 32;
  float* meta = s_meta;
  int* qkey = s_qkey + wv * kCandQueue;
  float* qx = s_qx + wv * kCandQueue;
  int* qcnt = s_qcnt + wv;
  if (SPARSE && lane == 0) *qcnt = 0;
  const unsigned colred_a = lds_addr(s_colred);
  const unsigned meta_a = lds_addr(meta), nmr_a = lds_addr(nmr_lds), qcnt_a = lds_addr(qcnt), qkey_a = lds_addr(qkey),
                 qx_a = lds_addr(qx);

  // One LDS-DMA instruction copies one 1 KiB fragment block (32 columns x one k-step x one lane half
  // pair) of the fragment-major planes: contiguous in global memory and in LDS.  Tile image in LDS:
  // [plane][column block 0/1][k-step][64 lanes x 16 B].
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
#pragma unroll
      for (int n = 0; n < INSTR_PER_WAVE; ++n) {
        const int blk = wv * INSTR_PER_WAVE + n;                       // 0 .. 2*KSTEPS-1 = (cb, ks)
        const _Float16* src = planes1[p] + ((long)(2 * t) * KSTEPS + blk) * 512 + lane * 8;
        glds16(src, smem + buf * BUF_BYTES + p * PLANE_BYTES + blk * 1024);
      }
    }
    // The tile's metadata travels the same way (one 4-byte LDS-DMA each, so no register carries a global
    // load across the sweep - hipcc would order every such load against the tile prefetch with vmcnt(0)):
    // wave 0 brings the 64 column stabilisers, wave 1 the 16 unit maxima of (wave, unit) = (lane/2, lane%2).
    if (MODE && wv == 0) glds4(a.nmc + (long)b * a.Sp + t * kTileCols + lane, meta + (buf & 1) * META);
    if (SPARSE && wv == 1 && lane < 16)
      glds4(a.umax + ((long)b * (a.Lp / 32) + panel * 8 + (lane >> 1)) * (a.Sp / 32) + 2 * t + (lane & 1),
            meta + (buf & 1) * META + 64);
  };
#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d)
    if (t0 + d < t1) stage(t0 + d, d);

  // ---- this wave's 32 rows as A fragments (lane (r,h): row r, k = h*C/2 + 8*ks + 0..7) ----
  // fragment-major planes (k_prep_split): one contiguous 1 KiB block per (32-row block, k-step);
  // lane = h*32 + r is exactly the lane order of that block.
  const int wrow0 = panel * kPanelRows + wv * 32;
  half8 ahi[KSTEPS], alo[MODE ? KSTEPS : 1];
  {
    const long off = (((long)b * a.Lp + wrow0) / 32 * KSTEPS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) ahi[ks] = *reinterpret_cast<const half8*>(a.hi0 + off + ks * 512);
    if (MODE) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) alo[ks] = *reinterpret_cast<const half8*>(a.lo0 + off + ks * 512);
    }
  }

  // row statistics, one per accumulator register (row = wrow0 + (g&3) + 8*(g>>2) + 4*h)
  float rstat[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat[g] = MODE ? 0.f : -INFINITY;
  // row stabilisers of this wave's 32 rows, parked in LDS (read back 4 at a time in the epilogue)
  float wmax_nmr = 0.f;     // largest (= least negative) row stabiliser of this wave's rows, uniform
  float emarg = 0.f;
  if (MODE) {
    const float nv = lane < 32 ? a.nmr[(long)b * a.Lp + wrow0 + lane] : -INFINITY;
    if (lane < 32) nmr_lds[lane] = nv;
    // padded rows (>= L) never contribute: keep them out of the wave's largest stabiliser
    if (SPARSE) { wmax_nmr = wave_max64(wrow0 + lane < a.L ? nv : -INFINITY); emarg = a.emarg[b]; }
  }
  const bool row_edge = (wrow0 + 32 > a.L);     // wave-uniform: some of this wave's rows are padding

  float* colout = a.colpart + ((long)b * a.panels + panel) * a.Sp;       // one partial per workgroup

  f32x16 acc;

  // one unit = 32 rows x 32 columns x C: the accumulator tile of this wave.
  // B fragments are read PF k-steps ahead into a small register ring so that the LDS latency
  // (~128 cycles) hides behind the MFMAs in between (96 cycles per k-step in the sum pass, 32 in the
  // max pass).  hipcc re-sinks plain LDS loads next to their use (one register pair, lgkmcnt(0) per
  // k-step), so the reads and their counted waits are inline asm; each wait names the fragments it
  // covers as "+v" operands, which keeps the MFMAs that consume them below it.  LDS returns in order,
  // so lgkmcnt(N) with N = reads issued after the needed ones is exact for our own reads and only
  // more conservative if the compiler has LDS/SMEM operations of its own in flight.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto mfma_unit = [&](int u) {
    const unsigned base = lds0 + (((u >> 1) - t0) % NBUF) * BUF_BYTES + (u & 1) * (KSTEPS * 1024) + lane * 16;
    constexpr int PF = MODE ? FM_PF_SUM : FM_PF_MAX;
    constexpr int RING = PF + 1;
    constexpr int RPK = MODE ? 2 : 1;       // LDS reads per k-step
    half8 bh[RING], bl[MODE ? RING : 1];
    auto issue = [&](int ks) {
      const unsigned la = base + (unsigned)(ks * 1024);
      asm volatile("ds_read_b128 %0, %1" : "=v"(bh[ks % RING]) : "v"(la));
      if (MODE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[ks % RING]) : "v"(la), "i"(PLANE_BYTES));
    };
#pragma unroll
    for (int ks = 0; ks < PF && ks < KSTEPS; ++ks) issue(ks);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + PF < KSTEPS) issue(ks + PF);
      const int ahead = (KSTEPS - 1 - ks) < PF ? (KSTEPS - 1 - ks) : PF;   // k-steps issued beyond ks
      if (MODE) {
        if (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]));
        else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]));
        else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]));
        else if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]));
        else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]));
      } else {
        if (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[ks % RING]));
        else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bh[ks % RING]));
        else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[ks % RING]));
        else if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bh[ks % RING]));
        else if (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[ks % RING]));
        else if (ahead == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(bh[ks % RING]));
        else if (ahead == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bh[ks % RING]));
        else if (ahead == 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(bh[ks % RING]));
        else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bh[ks % RING]));
      }
      static_assert(PF * RPK <= 8, "lgkmcnt ladder above covers at most 8 reads in flight");
      const half8 h8 = bh[ks % RING];
      if (ks == 0) {
        // The accumulator starts at 0, or at -inf for padded rows (>= L) / padded columns (>= S):
        // such entries then stay -inf through the whole chain, so the epilogue needs no masks
        // (max ignores them, exp2 gives 0, the candidate test fails).  Only edge waves pay for it.
        f32x16 z;
#pragma unroll
        for (int g = 0; g < 16; ++g) z[g] = 0.f;
        const int ucol0 = (u >> 1) * kTileCols + (u & 1) * 32;
        if (row_edge || ucol0 + 32 > a.S) {
          const float cb = (ucol0 + r < a.S) ? 0.f : -INFINITY;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = (wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h < a.L) ? cb : -INFINITY;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], h8, z, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], h8, acc, 0, 0, 0);
      }
      if (MODE) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bl[ks % RING], acc, 0, 0, 0);
      }
    }
  };

  // a candidate goes to its row's AND its column's slot list (k_select reads the row best from the one, the column
  // best from the other)
  auto record_candidate = [&](long grow, int col, float x) {
    const long gcol = (long)b * a.Sp + col;
    const int pos = atomicAdd(&cand_count[grow], 1);
    const int cpos = atomicAdd(&ccand_count[gcol], 1);
    if (pos < a.slots) { cand_j[grow * a.slots + pos] = col; cand_x[grow * a.slots + pos] = x; }
    if (cpos < a.slots) { ccand_i[gcol * a.slots + cpos] = (int)(grow - (long)b * a.Lp); ccand_x[gcol * a.slots + cpos] = x; }
    if (pos >= a.slots || cpos >= a.slots)
      atomicOr(a.flags, MODE == 1 ? (unsigned)FM_INT_SCREEN_OVERFLOW : (unsigned)FM_DEV_CANDIDATES);
  };

  // this lane's 16 row stabilisers (rows 8q + 4h + 0..3 of the wave's 32), four 16-byte LDS reads
  auto load_nmr = [&](f32x4 (&v)[4]) {
    const unsigned la = nmr_a + 16 * h;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(v[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(v[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(v[3]) : "v"(la));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  };

  // epilogue of unit u: fold the accumulator into the row / column statistics
  auto epilogue = [&](int u, float nmc) {
    const int col = (u >> 1) * kTileCols + (u & 1) * 32 + r;     // this lane's column
    const bool cvalid = col < a.S;
    float cstat;
    if (MODE == 0) {
      cstat = -INFINITY;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float x = acc[g];          // -inf for padded rows / columns (accumulator bias)
        rstat[g] = fmaxf(rstat[g], x);
        cstat = fmaxf(cstat, x);
      }
      cstat = combine_halves<false>(cstat);
      const float um = wave_max64(cstat);                         // block map for the sum pass
      if (lane == 0) a.umax[((long)b * (a.Lp / 32) + wrow0 / 32) * (a.Sp / 32) + u] = um;
    } else if (MODE == 3) {
      // dense conf_matrix (coarse_matching_new.py:68,70): softmax(sim,1) * softmax(sim,2) from the
      // log-softmax offsets nmr2 = nmr - log2(row sum), nmc2 = nmc - log2(column sum)
      cstat = 0.f;
      f32x4 nv4[4];
      load_nmr(nv4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float nm[4] = {nv4[q][0], nv4[q][1], nv4[q][2], nv4[q][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wrow0 + 8 * q + 4 * h + e;
          const float x = acc[4 * q + e];
          const float cf = __builtin_amdgcn_exp2f(__builtin_fmaf(x, kq, nm[e])) *
                           __builtin_amdgcn_exp2f(__builtin_fmaf(x, kq, nmc));
          if (row < a.L && cvalid) a.conf[((long)b * a.L + row) * a.S + col] = cf;
        }
      }
    } else {
      // largest min(row term, column term) per group of four accumulator registers (= rows 8q+4h .. +3):
      // tells the candidate scan below which groups to look at without recomputing anything
      float best4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      cstat = 0.f;
      float nmr[16];
      {
        f32x4 nv4[4];
        load_nmr(nv4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          nmr[4 * q] = nv4[q][0]; nmr[4 * q + 1] = nv4[q][1]; nmr[4 * q + 2] = nv4[q][2]; nmr[4 * q + 3] = nv4[q][3];
        }
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float x = acc[g];
        const float rr = __builtin_fmaf(x, kq, nmr[g]);   // x = -inf for padded rows / columns
        const float cc = __builtin_fmaf(x, kq, nmc);
        if (MODE == 1) {
          rstat[g] += __builtin_amdgcn_exp2f(rr);
          cstat += __builtin_amdgcn_exp2f(cc);
        }
        best4[g >> 2] = fmaxf(best4[g >> 2], fminf(rr, cc));
      }
      if (MODE == 1) cstat = combine_halves<true>(cstat);
      const float best = fmaxf(fmaxf(best4[0], best4[1]), fmaxf(best4[2], best4[3]));
#ifdef FM_ABL_NOCAND            // timing-only: no candidate recording
      if (false) {
#else
      if (__any(best > a.lt)) {      // some lane holds a candidate in this unit
#endif
#ifdef FM_DIAG_CLOCK
        ++diag_cand;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!__any(best4[q] > a.lt)) continue;                  // wave-uniform: usually 3 of the 4 groups
          int rbase = 8 * q + 4 * h;
          asm volatile("" : "+v"(rbase));   // keep the per-row values from being hoisted (and spilled)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int g = 4 * q + e;
            const float x = acc[g];
            const float rr = __builtin_fmaf(x, kq, nmr[g]);
            const float cc = __builtin_fmaf(x, kq, nmc);
            const int rl = rbase + e;                             // row inside this wave's 32
            if (rr > a.lt && cc > a.lt && wrow0 + rl < a.L && cvalid) {
              int qi;                                             // LDS atomic on the wave-private counter
              { const int one = 1;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(qi) : "v"(qcnt_a), "v"(one) : "memory"); }
              if (qi < kCandQueue) { lds_store_b32(qkey_a + qi * 4, (col << 5) | rl); lds_store_b32(qx_a + qi * 4, x * inv_sc); }
              else record_candidate((long)b * a.Lp + wrow0 + rl, col, x * inv_sc);     // queue full: straight to the lists
            }
          }
        }
      }
    }
    if (MODE <= 1 && h == 0)      // this wave's 32 rows of column `col`: parked for the fold after the tile barrier
      lds_store_b32(colred_a + ((((((u >> 1) - t0) & 1) * 8 + wv) * 64 + (u & 1) * 32 + r) * 4), cstat);
  };

  // Tile hand-over: LDS-DMA writes are ordered for other waves' reads only by the issuing wave's vmcnt
  // followed by a barrier.  A counted vmcnt leaves the younger tiles of the ring in flight across the
  // barrier (`__syncthreads()` would drain them: hipcc emits vmcnt(0) in front of it); VMEM returns in
  // order, so 'at most K outstanding' with K = instructions issued for the tiles after the wanted one
  // means the wanted tile has landed (stores in flight only make the wait more conservative).
  auto tile_barrier = [&](int tiles_after) {
    if (tiles_after <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (tiles_after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(GLDS_PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(2 * GLDS_PER_TILE) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  static_assert(NBUF <= 4 && 2 * GLDS_PER_TILE <= 63, "tile_barrier covers at most two tiles in flight");
  tile_barrier(min(NBUF - 2, t1 - t0 - 1));     // first tile and its metadata landed
  // hipcc does not see the counted waits above (inline asm) and would otherwise wait for everything loaded
  // before the loop at its first use INSIDE the loop - with `vmcnt(0)`, i.e. also for the tile prefetch just
  // issued there.  Naming the registers here makes it place that wait now, where everything has landed.
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" ::"v"(ahi[ks]));
  if (MODE) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" ::"v"(alo[ks]));
  }
  asm volatile("" ::"v"(emarg), "v"(wmax_nmr));
#ifdef FM_DIAG_CLOCK
  diag_pro = __builtin_amdgcn_s_memtime() - diag_c0;
#endif

  // (Waves w and w+4 share a SIMD; they drift apart by themselves, one in its MFMA chain while the other
  // runs its epilogue.  A forced one-unit stagger and s_setprio around the chain both measured slower once
  // the tile ring was truly asynchronous.)
#ifdef FM_ABL_NOTILES     // timing-only: prologue + final reduction, no sweep
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int u = 2 * t0; u < 2 * t0; ++u) {
#else
  for (int u = 2 * t0; u < 2 * t1; ++u) {
#endif
    const int t = u >> 1, par = (t - t0) & 1;
    if ((u & 1) == 0) {
      // refill the ring slot of tile t-1 (all waves left it at the previous barrier)
#ifndef FM_ABL_NOSTAGE   // timing-only ablations (results are wrong): never defined in the shipped build
      if (t + NBUF - 1 < t1) stage(t + NBUF - 1, (t - t0 + NBUF - 1) % NBUF);
#endif
    }

    float nmc_u = 0.f, um = 0.f;     // column stabiliser of this lane's column; largest raw f16 product of this unit
    if (MODE) {
      const unsigned ma = meta_a + (par * META + (u & 1) * 32 + r) * 4, mb = meta_a + (par * META + 64 + wv * 2 + (u & 1)) * 4;
      asm volatile("ds_read_b32 %0, %1" : "=v"(nmc_u) : "v"(ma));
      asm volatile("ds_read_b32 %0, %1" : "=v"(um) : "v"(mb));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nmc_u), "+v"(um));
    }
    bool skip = false;
    if (SPARSE) {
      const float top = __builtin_fmaf(um, kss, emarg);             // >= k * (exact product), log2 domain
      const float cmax = wave_max64(t * kTileCols + (u & 1) * 32 + r < a.S ? nmc_u : -INFINITY);
      skip = __builtin_amdgcn_readfirstlane((int)((top + wmax_nmr < -kSkipLog2) && (top + cmax < -kSkipLog2)));
    }
#ifdef FM_ABL_ALLSKIP     // timing-only: tile streaming and barriers, no unit work
    skip = true;
#endif
    if (!skip) {
#ifdef FM_DIAG_CLOCK
      ++diag_units;
#endif
      { DIAG_T0 mfma_unit(u); asm volatile("" :: "v"(acc)); DIAG_ADD(diag_mfma) }
#ifdef FM_ABL_NOEPI
      asm volatile("" :: "v"(acc));
#else
      { DIAG_T0 epilogue(u, nmc_u); DIAG_ADD(diag_epi) }
#endif
    } else if (MODE == 1 && h == 0) {
      lds_store_b32(colred_a + (((par * 8 + wv) * 64 + (u & 1) * 32 + r) * 4), 0.f);     // skipped unit: contributes nothing
    }
    if (u & 1) {
      // tile t consumed by every wave; tile t+1 landed; tiles t+2.. of the ring stay in flight
#ifndef FM_ABL_NOBAR
      { DIAG_T0 tile_barrier(min(t + NBUF - 1, t1 - 1) - (t + 1)); DIAG_ADD(diag_bar) }
#endif
      if (MODE <= 1 && wv == (t & 7)) {
        // every wave's column partials of tile t are in LDS (their writes precede the barrier): this wave folds
        // the 8 partials of the 64 columns in a fixed order (deterministic sums) and stores the workgroup's one
        float pv[8];
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8)
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(pv[w8]) : "v"(colred_a + (par * 8 * 64 + lane) * 4), "i"(w8 * 256));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]),
                     "+v"(pv[6]), "+v"(pv[7]));
        float cv = pv[0];
#pragma unroll
        for (int w8 = 1; w8 < 8; ++w8) cv = MODE ? cv + pv[w8] : fmaxf(cv, pv[w8]);
        if (MODE) colout[t * kTileCols + lane] = cv;
        else if (t * kTileCols + lane < a.S)      // max is exact and order independent: no partials, no reduction kernel
          __hip_atomic_fetch_max(a.colmax_u + (long)b * a.Sp + t * kTileCols + lane, ord_encode(cv), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef FM_DIAG_CLOCK
  diag_loop_end = __builtin_amdgcn_s_memtime();
#endif

  // Hand the parked candidates to the per-row slot lists: one entry per lane.  The slot reservation (a
  // global atomic with return, one memory round trip) is issued here and consumed after the row
  // reduction below, which hides most of its latency.
  int q_pos = -1, q_cpos = -1, q_key = 0;
  float q_x = 0.f;
  long q_row = 0, q_col = 0;
  if (SPARSE) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nq = min(*qcnt, kCandQueue);
    if (lane < nq) {
      q_key = qkey[lane];
      q_x = qx[lane];
      q_row = (long)b * a.Lp + wrow0 + (q_key & 31);
      q_col = (long)b * a.Sp + (q_key >> 5);
      q_pos = atomicAdd(&cand_count[q_row], 1);
      q_cpos = atomicAdd(&ccand_count[q_col], 1);
    }
  }
  auto commit_candidates = [&]() {
    if (SPARSE && q_pos >= 0) {
      if (q_pos < a.slots) { cand_j[q_row * a.slots + q_pos] = q_key >> 5; cand_x[q_row * a.slots + q_pos] = q_x; }
      if (q_cpos < a.slots) { ccand_i[q_col * a.slots + q_cpos] = wrow0 + (q_key & 31); ccand_x[q_col * a.slots + q_cpos] = q_x; }
      if (q_pos >= a.slots || q_cpos >= a.slots)
        atomicOr(a.flags, MODE == 1 ? (unsigned)FM_INT_SCREEN_OVERFLOW : (unsigned)FM_DEV_CANDIDATES);
    }
  };
  if (MODE >= 2) { commit_candidates(); return; }
  // ---- row statistics of this workgroup's column range: reduce over the 32 lanes of each half ----
  // (DPP butterflies: the former ds_bpermute shuffles cost ~3.7k cycles of LDS round trips per wave)
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat[g] = half_reduce32<MODE != 0>(rstat[g]);
  if (r == 0) {
    if (MODE) {
      float* out = a.rowpart + ((long)b * a.splits + split) * a.Lp + wrow0 + 4 * h;
#pragma unroll
      for (int g = 0; g < 16; ++g) out[(g & 3) + 8 * (g >> 2)] = rstat[g];
    } else {
      unsigned* out = a.rowmax_u + (long)b * a.Lp + wrow0 + 4 * h;
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (wrow0 + 4 * h + (g & 3) + 8 * (g >> 2) < a.L)
          __hip_atomic_fetch_max(out + (g & 3) + 8 * (g >> 2), ord_encode(rstat[g]), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  commit_candidates();
#ifdef FM_DIAG_CLOCK       // candidate slots of the padded rows (>= L, never used) carry the stamps of 64 waves
  if (MODE <= 1 && split == 0 && panel < 8 && (a.Lp - a.L) * a.slots >= 512 && lane < 8) {
    const float vals[8] = {(float)(__builtin_amdgcn_s_memtime() - diag_c0),
                           (float)(__builtin_amdgcn_s_memrealtime() - diag_r0), (float)diag_units, (float)diag_mfma,
                           (float)diag_epi, (float)diag_bar, (float)diag_pro,
                           (float)(__builtin_amdgcn_s_memtime() - diag_loop_end)};
    float vv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) vv = lane == q ? vals[q] : vv;
    a.cand_x[((long)b * a.Lp + a.L) * a.slots + (panel * 8 + wv) * 8 + lane] = vv;
  }
#endif
}

template <int C, int MODE>
static hipError_t launch_corr_t(const CorrArgs& a, int blocks, hipStream_t st) {
  constexpr int BUF_BYTES = (MODE ? 2 : 1) * kTileCols * C * 2;
  constexpr int SMEM = (MODE ? 2 : FM_NBUF_MAX) * BUF_BYTES;      // the tile ring; the small tables are static LDS
  static unsigned long long lds_set = 0;      // one flag word per template instance
  hipError_t e = ensure_dynamic_lds(&k_corr<C, MODE>, SMEM, &lds_set);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_corr<C, MODE>), dim3(blocks), dim3(512), SMEM, st, a);
  return hipGetLastError();
}

hipError_t launch_corr(int mode, const CoarseWs& w, char* base, float inv_ct, float thr, hipStream_t st, float* conf) {
  CorrArgs a;
  a.hi0 = (const _Float16*)(base + w.hi0); a.lo0 = (const _Float16*)(base + w.lo0);
  a.hi1 = (const _Float16*)(base + w.hi1); a.lo1 = (const _Float16*)(base + w.lo1);
  a.nmr = (const float*)(base + (mode >= 2 ? w.nmr2 : w.nmr));
  a.nmc = (const float*)(base + (mode >= 2 ? w.nmc2 : w.nmc));
  a.conf = conf;
  a.rowpart = (float*)(base + w.rowB);
  a.colpart = (float*)(base + w.colB);
  a.rowmax_u = (unsigned*)(base + w.rowmax_u); a.colmax_u = (unsigned*)(base + w.colmax_u);
  a.dense_cnt = (const int*)(base + w.dense_cnt);
  a.cand_count_b = (int*)(base + w.cand_count_b); a.cand_j_b = (int*)(base + w.cand_j_b);
  a.cand_x_b = (float*)(base + w.cand_x_b);
  a.ccand_count = (int*)(base + w.ccand_count); a.ccand_i = (int*)(base + w.ccand_i); a.ccand_x = (float*)(base + w.ccand_x);
  a.ccand_count_b = (int*)(base + w.ccand_count_b); a.ccand_i_b = (int*)(base + w.ccand_i_b);
  a.ccand_x_b = (float*)(base + w.ccand_x_b);
  a.sigimg = (const float*)(base + w.sigimg);
  a.dense_units = &((const Scalars*)(base + w.scalars))->dense_units;
  a.umax = (float*)(base + w.umax); a.emarg = (const float*)(base + w.emarg);
  a.f16inv = (const float*)(base + w.f16inv);
  a.cand_count = (int*)(base + w.cand_count); a.cand_j = (int*)(base + w.cand_j);
  a.cand_x = (float*)(base + w.cand_x);
  a.flags = (unsigned*)(base + w.scalars);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.tiles = w.tiles;
  a.splits = mode ? w.splits : w.splits0; a.tiles_per_split = (w.tiles + a.splits - 1) / a.splits; a.slots = w.slots;
  {
    // one XCD runs ~blocks/8 workgroups: make its block of (panels x splits) as square as the bytes are
    // (a panel contributes 256 rows of A, a split tiles_per_split*64 columns of B)
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.tiles_per_split * kTileCols) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
    if (a.splits == 1) a.pgroup = w.panels;
#ifdef FM_TUNE_ENV
    if (const char* e = getenv("FM_PGROUP")) a.pgroup = atoi(e) < 1 ? 1 : (atoi(e) > w.panels ? w.panels : atoi(e));
#endif
  }
  a.k = inv_ct * kLog2e;
  a.lt = log2f(thr) - (mode == 2 ? 2e-4f : 0.f);   // pass C compares rounded log-softmax values: small guard
  const int blocks = w.N * a.splits * w.panels;
#define FM_CORR_CASE(CC)                                                     \
  case CC: return mode == 2 ? launch_corr_t<CC, 2>(a, blocks, st)            \
                 : hipErrorInvalidValue;       /* mode 1 (the dense sum sweep) and mode 3 (the dense conf_matrix) are \
                                                  k_dense / k_dense<C, CONF>, coarse_dense.hip */
  switch (w.C) {
    FM_CORR_CASE(64)
    FM_CORR_CASE(128)
    FM_CORR_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_CORR_CASE
}

}  // namespace fm
