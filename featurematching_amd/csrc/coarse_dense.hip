// Coarse stage, the DENSE sum kernel (round 4): dual-softmax denominators and candidates of the samples whose similarity
// is flat - every entry matters - from a float32-equivalent product on the f16 matrix cores.
//
// Reproduces network/utils/coarse_matching_new.py:64-68 for the samples the screening kernel flagged (dense_cnt[b] > 0):
//   raw = hi0.hi1 + lo0.hi1 + hi0.lo1   (float16 hi / lo planes of k_prep_f16: 22 significant bits)
//   row sums  sum_j exp2(k raw - m^_i), column sums  sum_i exp2(k raw - c^_j)  with the stabilisers of the screening
//   kernel, and the candidate list {(i,j): both terms > thr} - a superset of every entry with conf > thr (:99).
// It replaces k_corr<C,1> of rounds 1-3, whose in-kernel stamps at 640x480 read: 85k cycles per wave for 12 units of
// 1.5k cycles of matrix work each - 10k of prologue, 2.1k + 1.1k per unit for the MFMA chain and the epilogue one after
// the other, 9k in barriers of a two-deep tile ring, 2k per unit of bookkeeping (unit maxima, skip tests, wave-wide
// maxima).  What is different here:
//   * NO block-sparse skipping: a sample comes here because its similarity is flat; the unit maxima, their margins and
//     the per-unit wave-wide reductions are gone;
//   * the ring holds four UNITS (32 columns, hi + lo = 32 KiB at C = 256) instead of two 64-column tiles: three units
//     of LDS-DMA in flight, handed over by counted vmcnt + one raw s_barrier per unit;
//   * the epilogue of unit u-1 (exp2 of the row and of the column term of 16 accumulator registers, the running sums,
//     the candidate test) is sliced over the 16 k-steps of unit u's MFMA chain: a wave issues in order, so the vector
//     work only runs under the matrix cores' 96 cycles per k-step if it sits between the MFMAs in program order;
//   * the rows' stabilisers come from one register by v_readlane (no LDS reads between the counted B-fragment waits).
// Structure otherwise as before: one workgroup = 8 waves = a 256-row panel x a range of columns; each wave keeps its 32
// rows' hi and lo A fragments (128 VGPRs at C = 256) for the whole sweep; the 8 waves' column sums of a unit meet in
// LDS and one wave folds them in a fixed order (deterministic) into the panel's partial.
#include <type_traits>

#include "fm_device.h"

namespace fm {

typedef _Float16 half8d __attribute__((ext_vector_type(8)));
typedef float f32x16d __attribute__((ext_vector_type(16)));

constexpr int kDenseQueue = 64;       // candidates a wave parks in LDS (one per lane at the hand-over)
constexpr int kDenseRing = 4;         // units of the LDS ring

struct DenseArgs {
  const _Float16* hi0; const _Float16* lo0; const _Float16* hi1; const _Float16* lo1;
  const float* nmr; const float* nmc;
  float* rowpart; float* colpart;       // [N][splits][Lp], [N][panels][Sp]
  const int* dense_cnt; const int* dense_units; const float* f16inv;
  int* cand_count; int* cand_j; float* cand_x; int* ccand_count; int* ccand_i; float* ccand_x;   // the dense kernel's lists
  unsigned* flags;
  float* diag;                          // diagnostic build: stamp buffer (the screening kernel's cand_x region)
  float* conf;                          // CONF variant: the dense [N, L, S] conf_matrix (nmr / nmc then hold the log-softmax offsets)
  int L, S, Lp, Sp, panels, units, splits, units_per_split, slots, pgroup;
  float k, lt;
};

__device__ __forceinline__ int xcd_remap_d(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov_d(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xf, BANK, false));
}
// sum over the 32 lanes that share lane >> 5, result in every lane of the half; DPP only (fixed order)
__device__ __forceinline__ float half_sum32_d(float v) {
  v = v + dpp_mov_d<0xB1, 0xf>(v, v);
  v = v + dpp_mov_d<0x4E, 0xf>(v, v);
  { float t = dpp_mov_d<0x104, 0x5>(v, v); t = dpp_mov_d<0x114, 0xA>(t, v); v = v + t; }
  v = v + dpp_mov_d<0x128, 0xf>(v, v);
  { float p = v, q = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q)); v = p + q; }
  return v;
}
__device__ __forceinline__ float halves_sum_d(float v) {      // lane l + lane l ^ 32
  float p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return p + q;
}
template <typename T>
__device__ __forceinline__ unsigned lds_addr_d(T* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) T*)p; }

// CONF (round 4): the same sweep writes data['conf_matrix'] (coarse_matching_new.py:68,70) for EVERY sample instead of
// forming sums and candidates - conf = exp2(k x + nmr2_i) * exp2(k x + nmc2_j) with the log-softmax offsets of
// k_reduce_sums, the arithmetic of k_corr<C,3>, which it replaces (2.50 ms for the 5.9 GB of a 64-pair batch: a two-deep
// tile ring whose epilogue - and with it every store - ran behind the MFMA chain).  The epilogue slice of a register is
// two fused multiply-adds, two exp2, one multiply and ONE store (a scalar row base + a per-lane offset: 32 consecutive
// floats of two rows per instruction), spread over the next unit's k-steps like the sums' slices.
// RESCREEN (round 4; replaces k_corr<C,2>, the last user of the tile-based sweep of rounds 1-3): the exact re-screening
// of FM_MODE_EXACT_SCREENING - only when the sum sweep overflowed a row's or a column's candidate slots: the same
// product again, the candidate test with the now-known softmax denominators (log2 P_row > log2 thr and log2 P_col >
// log2 thr: at most 1/thr entries of a row can pass), no sums; the lists of the samples the sum sweep served are
// refilled from scratch (k_reduce_sums cleared their counters).  Enqueued always, exits at once when not needed.
// CONF_LITE (round 5): the conf sweep of the samples the SCREENING kernel served.  There every entry that is not on its
// row's list of significant entries has exp2(k x + nm) < 2^-32 of its row's largest term - conf < 2.4e-10 whatever a few
// per cent of relative error do to it - and every entry that is on a list is rewritten from its exact float32 dot product
// by k_conf_patch: ONE float16 product (hi x hi, 11 bits) is enough.  A third of the MFMAs and, more to the point, half
// of the plane bytes per unit: the sweep is bound by its memory stream (tools/microbench_store_pattern.hip: the same
// stores with 32 KiB of plane reads per unit and NO arithmetic take 1.6-1.9 ms at cfg#3, with reads that always hit 0.93).
// The samples the dense kernel served (flat similarity: entries of every size) keep the hi/lo-split product (CONF).
enum { kDenseSums = 0, kDenseConf = 1, kDenseRescreen = 2, kDenseConfLite = 3 };
template <int C, int VAR = kDenseSums>
__global__ __launch_bounds__(512) void k_dense(DenseArgs a) {
  constexpr bool LITE = VAR == kDenseConfLite;
  constexpr bool CONF = VAR == kDenseConf || LITE, RESCREEN = VAR == kDenseRescreen, SUMS = VAR == kDenseSums;
  constexpr int KSTEPS = C / 16;
  constexpr int PLANE = KSTEPS * 1024;              // bytes of one plane (hi or lo) of a unit
  constexpr int UNIT_BYTES = 2 * PLANE;             // hi then lo
  constexpr int NPIECE = LITE ? KSTEPS : 2 * KSTEPS;      // LDS-DMA pieces (1 KiB) per unit: (plane, k-step)
  constexpr int PW = (NPIECE + 7) / 8;              // ... per wave (LITE at C = 64: 4 pieces - waves 4 .. 7 bring none)
  constexpr int SPK = 16 / KSTEPS;                  // epilogue slices (accumulator registers) per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];      // the unit ring
  __shared__ float s_meta[kDenseRing * 64];         // per ring slot: the unit's 32 column stabilisers (twice)
  __shared__ float s_colred[3 * 8 * 32];            // per unit (mod 3): the 8 waves' column sums of 32 columns
  __shared__ int s_qkey[8 * kDenseQueue];           // wave-private candidate queue: (col << 5) | local row
  __shared__ float s_qx[8 * kDenseQueue];

  if (SUMS && *a.dense_units == 0) return;          // uniform: the screening kernel handled every sample
  if (RESCREEN && !(*a.flags & FM_INT_SCREEN_OVERFLOW)) return;     // uniform: the sum sweep's screening sufficed
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps per phase of every wave (tools/diag_clock.py)
  const unsigned long long dg0 = __builtin_amdgcn_s_memtime(), dgr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long dg_chain = 0, dg_bar = 0, dg_pro = 0, dg_loop_end = 0;
#define DG_T0 const unsigned long long dg_t = __builtin_amdgcn_s_memtime();
#define DG_ADD(x) x += __builtin_amdgcn_s_memtime() - dg_t;
#else
#define DG_T0
#define DG_ADD(x)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int kk = xcd_remap_d(blockIdx.x, gridDim.x);
  const int per_sample = a.panels * a.splits;
  const int b = kk / per_sample;
  kk -= b * per_sample;
  const int gsz = a.pgroup * a.splits;
  const int pg = kk / gsz;
  kk -= pg * gsz;
  const int pcount = min(a.pgroup, a.panels - pg * a.pgroup);
  const int split = kk / pcount;
  const int panel = pg * a.pgroup + (kk - split * pcount);
  const int u0 = split * a.units_per_split, u1 = min(u0 + a.units_per_split, a.units);
  if (!CONF && a.dense_cnt[b] == 0) return;         // uniform: this sample was served by the screening kernel
  if (CONF && LITE != (a.dense_cnt[b] == 0)) return;        // uniform: the conf sweep of this sample is the other variant's
  const float inv_sc = a.f16inv[b];                 // the planes carry exact power-of-two scales (k_prep_f16)
  const float kq = a.k * inv_sc;                    // accumulator -> log2-domain similarity
  const char* plane_hi = reinterpret_cast<const char*>(a.hi1 + (long)b * a.Sp * C);
  const char* plane_lo = reinterpret_cast<const char*>(a.lo1 + (long)b * a.Sp * C);
  const unsigned ring_a = lds_addr_d(smem), meta_a = lds_addr_d(s_meta), colred_a = lds_addr_d(s_colred);
  const unsigned qkey_a = lds_addr_d(s_qkey + wv * kDenseQueue), qx_a = lds_addr_d(s_qx + wv * kDenseQueue);

  // one unit = 2 * KSTEPS pieces of 1 KiB ((plane, k-step): the 64 lanes of an MFMA operand fragment), contiguous in
  // the fragment-major planes; the waves take the pieces round robin; wave 0 also brings the unit's 32 column stabilisers
  auto stage = [&](int u) {
    const int slot = (u - u0) % kDenseRing;
#pragma unroll
    for (int n = 0; n < PW; ++n) {
      const int p = wv * PW + n;                    // 0 .. 2 KSTEPS - 1: hi pieces first, then lo
      if (NPIECE < 8 * PW && p >= NPIECE) break;
      const char* src = (p < KSTEPS ? plane_hi : plane_lo) + ((long)u * KSTEPS + (p < KSTEPS ? p : p - KSTEPS)) * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + slot * UNIT_BYTES + p * 1024), 16, 0, 0);
    }
    if (wv == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.nmc + (long)b * a.Sp + u * 32 + r),
                                       (__attribute__((address_space(3))) void*)(s_meta + slot * 64), 4, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < kDenseRing - 1; ++d)
    if (u0 + d < u1) stage(u0 + d);

  // ---- this wave's 32 rows as A fragments (lane (r,h): row r, k = h*C/2 + 8*ks + 0..7), hi and lo ----
  const int wrow0 = panel * kPanelRows + wv * 32;
  // -stabiliser*log2e of row wrow0 + (lane & 31) (both halves hold it), requested AHEAD of the A fragments: older in the
  // vmcnt order, so picking the rows' values out of it below does not wait for them
  const float nv = a.nmr[(long)b * a.Lp + wrow0 + r];
  half8d ahi[KSTEPS], alo[KSTEPS];
  {
    const long off = (((long)b * a.Lp + wrow0) / 32 * KSTEPS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      ahi[ks] = *reinterpret_cast<const half8d*>(a.hi0 + off + ks * 512);
      if constexpr (!LITE) alo[ks] = *reinterpret_cast<const half8d*>(a.lo0 + off + ks * 512);
    }
  }
  float nmsel[16];                                  // ... of this lane's 16 rows (accumulator register g: row (g&3) + 8 (g>>2) + 4 h)
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row0 = (g & 3) + 8 * (g >> 2);
    const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nv), row0));
    const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nv), row0 + 4));
    nmsel[g] = h ? n1 : n0;
  }
  const bool row_edge = (wrow0 + 32 > a.L);
  int qn = 0;                                       // parked candidates (per lane: the wave's count is kept in lane 0's view)
  float rstat[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat[g] = 0.f;

  auto unit_barrier = [&](int units_after) {        // the next unit has landed; later ones stay in flight
    if (units_after <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (units_after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  // The first unit has landed once its pieces - the OLDEST vector-memory operations of this wave - are done: everything
  // younger (the next units' pieces, the 2 KSTEPS A-fragment loads, the stabilisers) may stay in flight across the
  // barrier; the first chain then takes each A fragment as it arrives (the compiler's own counted waits), instead of
  // the whole prologue - ~350 KB per workgroup through one CU's 64 B/clk - in front of the first MFMA (12k cycles).
  {
    constexpr int YOUNGER = LITE ? KSTEPS : 2 * KSTEPS;      // the A fragments (the other units' pieces only make it safer)
    static_assert(YOUNGER <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(YOUNGER) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  float* colout = a.colpart + ((long)b * a.panels + panel) * a.Sp;
  auto fold_columns = [&](int u) {                  // the 8 waves' sums of unit u's 32 columns, in wave order
    if (!SUMS || wv != (u & 7) || lane >= 32) return;
    const unsigned ad = colred_a + (((u - u0) % 3) * 8 * 32 + lane) * 4;
    float pv[8];
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(pv[w8]) : "v"(ad), "i"(w8 * 128));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]),
                 "+v"(pv[6]), "+v"(pv[7]));
    float cv = pv[0];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) cv += pv[w8];
    colout[u * 32 + lane] = cv;
  };

  f32x16d accC;                                     // the accumulators of the unit whose epilogue is due
  float nmc_c = 0.f;                                // its column stabiliser (this lane's column)
#pragma unroll
  for (int g = 0; g < 16; ++g) accC[g] = 0.f;

  // one pass: the MFMA chain of unit `un` (when DN) with the epilogue of unit `uc` (when DC) sliced between its k-steps
  auto pass = [&](auto dn, auto dc, int un, int uc) {
    constexpr bool DN = decltype(dn)::value, DC = decltype(dc)::value;
    f32x16d accN;
    float nmc_n = 0.f;
    float cstat = 0.f;
    unsigned bm = 0;                                  // this lane's candidate registers of unit uc: bit 15 - g
    const float kqv = kq, ltv = a.lt;
    // CONF: unit uc's tile of the matrix - a scalar base (row wrow0, column 32 uc), the lane's own offset (row 4 h, column
    // r), the row pitch in bytes; a (wave, unit) that touches the matrix's edge stores behind the chain, predicated
    const char* conf_base = nullptr;
    long conf_pitch = 0;
    unsigned conf_voff = 0;
    bool conf_full = false;
    if constexpr (CONF) {
      if (DC) {
        conf_pitch = (long)a.S * 4;
        conf_base = reinterpret_cast<const char*>(a.conf + ((long)b * a.L + wrow0) * a.S + (long)uc * 32);
        conf_voff = (unsigned)((4 * h * a.S + r) * 4);
        conf_full = !row_edge && uc * 32 + 32 <= a.S;
#ifdef FM_ABL_CONF_SMALLDST     // timing-only ablation (results are wrong): every store lands in the same 64 KiB
        conf_pitch = 0;
        conf_base = reinterpret_cast<const char*>(a.conf) + wv * 8192;
        conf_voff = (unsigned)(lane * 4);
#endif
      }
    }
    const unsigned base = ring_a + ((un - u0) % kDenseRing) * UNIT_BYTES + lane * 16;
    constexpr int PF = KSTEPS < 3 ? KSTEPS - 1 : 2;      // B-fragment read-ahead (k-steps): LDS latency under 8 waves' reads
    constexpr int RING = PF + 1;                         // is ~300 cycles, a k-step of this wave ~150
    half8d bh[RING], bl[RING];
    // (a capture-less lambda with explicit operands: hipcc does not let a generic lambda nested in a generic lambda
    // capture the enclosing one's arrays)
    auto issue_ = [](auto ksc, half8d (&bh_)[RING], half8d (&bl_)[RING], unsigned base_) {
      constexpr int ks = decltype(ksc)::value;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh_[ks % RING]) : "v"(base_), "i"(ks * 1024));
      if constexpr (!LITE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl_[ks % RING]) : "v"(base_), "i"(PLANE + ks * 1024));
    };
#define issue(...) issue_(__VA_ARGS__, bh, bl, base)
    if (DN) {
      asm volatile("ds_read_b32 %0, %1" : "=v"(nmc_n) : "v"(meta_a + (((un - u0) % kDenseRing) * 64 + r) * 4));
      issue(std::integral_constant<int, 0>{});
      if constexpr (PF > 1) issue(std::integral_constant<int, 1>{});
    }
    auto kstep = [&](auto self, auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if (DN) {
        if constexpr (ks + PF < KSTEPS) issue(std::integral_constant<int, ks + PF>{});
        constexpr int ahead = (KSTEPS - 1 - ks) < PF ? (KSTEPS - 1 - ks) : PF;     // k-steps issued beyond ks
        if constexpr (LITE) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[ks % RING]), "+v"(nmc_n) : "n"(ahead));
        else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(bh[ks % RING]), "+v"(bl[ks % RING]), "+v"(nmc_n) : "n"(2 * ahead));
        if (ks == 0) {
          // The accumulator starts at 0, or at -inf for padded rows (>= L) / padded columns (>= S): such entries stay
          // -inf through the whole chain, so the epilogue needs no masks (exp2 gives 0, the candidate test fails)
          f32x16d z;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = 0.f;
          if (row_edge || un * 32 + 32 > a.S) {
            const float cb = (un * 32 + r < a.S) ? 0.f : -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g) z[g] = (wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h < a.L) ? cb : -INFINITY;
          }
          accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bh[ks % RING], z, 0, 0, 0);
        } else {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bh[ks % RING], accN, 0, 0, 0);
        }
#ifndef FM_ABL_DENSE_ONEMFMA    // timing-only ablation: one product instead of three
        if constexpr (!LITE) {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], bh[ks % RING], accN, 0, 0, 0);
          accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bl[ks % RING], accN, 0, 0, 0);
        }
#endif
      }
#ifndef FM_ABL_DENSE_NOEPI      // timing-only ablation (results are wrong): never defined in the shipped build
      if (DC) {
#pragma unroll
        for (int sg = 0; sg < SPK; ++sg) {
          const int g = ks * SPK + sg;
          // One slice = one accumulator register (x = -inf for padded rows / columns), as volatile asm: hipcc moved the
          // row half of a C++ version behind the whole MFMA chain, where nothing hides it.  Seven vector instructions:
          // the SIMD's instruction issue is what two waves' epilogues and MFMAs compete for (a 15-instruction slice
          // - stabiliser by v_readlane + per-half exec masks, min / max candidate filter - cost 35k of 94k cycles).
          // Candidates (both terms > thr) are marked on the spot: v_cmp + v_addc shift the hit into a per-lane bit mask
          // (bit 15 - g).  A filter + rescan of the 16 registers took ~1k cycles whenever ANY of the workgroup's 8 waves
          // entered it - nearly every unit - and the per-unit barrier made all of them wait: 13k of 82k cycles.
          if constexpr (CONF) {
            // (round 5, in-kernel clock at cfg#3, tools/diag_conf_clock.py: 6140 cycles per unit and workgroup; the same
            // kernel with the store removed 3160 at one product (FM_ABL_CONF_NOSTORE), with every store aimed at the same
            // 64 KiB 3350 (FM_ABL_CONF_SMALLDST): what the stores cost is the write path behind the L2, ~160 cycles of
            // wave time each, while the whole sweep writes at 2.5 TB/s - a store-only kernel of the same shape and
            // occupancy reaches 5.5 (tools/microbench_store_pattern.hip).  Tried without effect: counting the stores in
            // the per-unit vmcnt wait instead of draining them, the SIMD's two waves storing in alternate halves of the
            // k-steps, the conf value formed in the accumulator register (no write-after-read on the store's data),
            // nontemporal stores (slower there: 7070).)
            // conf of register g's entry and its store: rows (g & 3) + 8 (g >> 2) [+ 4 for the upper half], column r
            float t1, t2;
            const char* rowbase = conf_base + (long)((g & 3) + 8 * (g >> 2)) * conf_pitch;      // (wave-uniform: scalar)
            if (conf_full) {
              // (nontemporal stores in the one-product sweep, whose limit is the memory stream: 3.05 -> 2.84 ms for the whole
              // conf stage at cfg#3; in the hi/lo-split sweep, which the matrix cores' power budget clocks down, they cost time:
              // 7070 against 6410 cycles per unit)
#if defined(FM_ABL_CONF_NOEXP)           // timing-only ablations (results are wrong): never defined in the shipped build
#define FM_CONF_EXP ""
#else
#define FM_CONF_EXP "v_exp_f32 %[t1], %[t1]\n\tv_exp_f32 %[t2], %[t2]\n\ts_nop 0\n\t"
#endif
#if defined(FM_ABL_CONF_NOSTORE)
#define FM_CONF_STORE(POLICY) ""
#else
#define FM_CONF_STORE(POLICY) "global_store_dword %[vo], %[t1], %[sb]" POLICY
#endif
#define FM_CONF_SLICE(POLICY)                                                                                              \
  asm volatile("v_fma_f32 %[t1], %[x], %[kq], %[nm]\n\t"                                                                   \
               "v_fma_f32 %[t2], %[x], %[kq], %[nmc]\n\t" FM_CONF_EXP "v_mul_f32 %[t1], %[t1], %[t2]\n\t" FM_CONF_STORE(POLICY) \
               : [t1] "=&v"(t1), [t2] "=&v"(t2)                                                                            \
               : [x] "v"(accC[g]), [kq] "v"(kqv), [nm] "v"(nmsel[g]), [nmc] "v"(nmc_c), [vo] "v"(conf_voff), [sb] "s"(rowbase) \
               : "memory")
              if constexpr (LITE) FM_CONF_SLICE(" nt");
              else FM_CONF_SLICE("");
#undef FM_CONF_SLICE
#undef FM_CONF_STORE
#undef FM_CONF_EXP
            }
          } else if constexpr (RESCREEN) {
            // log2 P_row and log2 P_col of register g's entry against log2 thr: the hit goes into the lane's bit mask
            float t1, t2;
            asm volatile(
                "v_fma_f32 %[t1], %[x], %[kq], %[nm]\n\t"
                "v_fma_f32 %[t2], %[x], %[kq], %[nmc]\n\t"
                "v_min_f32 %[t1], %[t1], %[t2]\n\t"
                "v_cmp_lt_f32 vcc, %[lt], %[t1]\n\t"
                "v_addc_co_u32 %[bm], vcc, %[bm], %[bm], vcc"
                : [t1] "=&v"(t1), [t2] "=&v"(t2), [bm] "+v"(bm)
                : [x] "v"(accC[g]), [kq] "v"(kqv), [nm] "v"(nmsel[g]), [nmc] "v"(nmc_c), [lt] "v"(ltv)
                : "vcc");
          } else {
          float t1, t2, t3;
          asm volatile(
              "v_fma_f32 %[t1], %[x], %[kq], %[nm]\n\t"
              "v_fma_f32 %[t2], %[x], %[kq], %[nmc]\n\t"
              "v_min_f32 %[t3], %[t1], %[t2]\n\t"
#ifndef FM_ABL_DENSE_NOEXP       // timing-only ablation
              "v_exp_f32 %[t1], %[t1]\n\t"
              "v_exp_f32 %[t2], %[t2]\n\t"
#endif
              "v_cmp_lt_f32 vcc, %[lt], %[t3]\n\t"
              "v_addc_co_u32 %[bm], vcc, %[bm], %[bm], vcc\n\t"
              "v_add_f32 %[rs], %[rs], %[t1]\n\t"
              "v_add_f32 %[cs], %[cs], %[t2]"
              : [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [bm] "+v"(bm), [rs] "+v"(rstat[g]), [cs] "+v"(cstat)
              : [x] "v"(accC[g]), [kq] "v"(kqv), [nm] "v"(nmsel[g]), [nmc] "v"(nmc_c), [lt] "v"(ltv)
              : "vcc");
          }
        }
      }
#endif
      if (DN && DC) __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + 1 < KSTEPS) self(self, std::integral_constant<int, ks + 1>{});
    };
    kstep(kstep, std::integral_constant<int, 0>{});
#undef issue
    if constexpr (CONF) {
      if (DC && !conf_full && wrow0 < a.L) {        // the matrix's edge: predicated stores (x = -inf beyond it)
        const int col = uc * 32 + r;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h;
          const float cf = __builtin_amdgcn_exp2f(__builtin_fmaf(accC[g], kqv, nmsel[g])) *
                           __builtin_amdgcn_exp2f(__builtin_fmaf(accC[g], kqv, nmc_c));
          if (row < a.L && col < a.S) a.conf[((long)b * a.L + row) * a.S + col] = cf;
        }
      }
    }
    if (DC && !CONF) {
      if constexpr (SUMS) {
        cstat = halves_sum_d(cstat);                // this wave's 32 rows of column r
        if (h == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(colred_a + ((((uc - u0) % 3) * 8 + wv) * 32 + r) * 4), "v"(cstat) : "memory");
      }
      {
        // (padded rows / columns carry x = -inf: never marked)
        unsigned long long hitl = __ballot(bm != 0);
        while (hitl) {                              // wave-uniform, usually one lane
          const int l = __builtin_ctzll(hitl);
          hitl &= hitl - 1;
          unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)bm, l);
          while (bits) {
            const int g = 15 - __builtin_ctz(bits);
            bits &= bits - 1;
            const int key = ((uc * 32 + (l & 31)) << 5) | ((g & 3) + 8 * (g >> 2) + 4 * (l >> 5));
            // (accumulator register g of lane l: a dynamic register index - sixteen selects, on this rare path)
            float xg = accC[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) xg = g == q ? accC[q] : xg;
            const float xv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xg), l)) * inv_sc;
            if (qn < kDenseQueue) {
              if (lane == 0) {
                asm volatile("ds_write_b32 %0, %1" ::"v"(qkey_a + qn * 4), "v"(key) : "memory");
                asm volatile("ds_write_b32 %0, %1" ::"v"(qx_a + qn * 4), "v"(xv) : "memory");
              }
            } else if (lane == 0) {                 // queue full: straight to the lists
              const long grow = (long)b * a.Lp + wrow0 + (key & 31), gcol = (long)b * a.Sp + (key >> 5);
              const int pos = atomicAdd(&a.cand_count[grow], 1), cpos = atomicAdd(&a.ccand_count[gcol], 1);
              if (pos < a.slots) { a.cand_j[grow * a.slots + pos] = key >> 5; a.cand_x[grow * a.slots + pos] = xv; }
              if (cpos < a.slots) { a.ccand_i[gcol * a.slots + cpos] = wrow0 + (key & 31); a.ccand_x[gcol * a.slots + cpos] = xv; }
              if (pos >= a.slots || cpos >= a.slots) atomicOr(a.flags, RESCREEN ? (unsigned)FM_DEV_CANDIDATES : (unsigned)FM_INT_SCREEN_OVERFLOW);
            }
            ++qn;
          }
        }
      }
    }
    if (DN) {
#pragma unroll
      for (int g = 0; g < 16; ++g) accC[g] = accN[g];
      nmc_c = nmc_n;
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  // Per unit u: refill the ring slot of unit u-1 (every wave left it at the previous barrier) | MFMAs(u) with the
  // epilogue of u-1 between them | barrier: every wave has read unit u, unit u+1 has landed | fold unit u-1's columns
#ifdef FM_DIAG_CLOCK
  dg_pro = __builtin_amdgcn_s_memtime() - dg0;
#endif
#ifdef FM_DENSE_PRIO      // experiment: static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (wv >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  if (u0 < u1) {
    if (u0 + kDenseRing - 1 < u1) stage(u0 + kDenseRing - 1);
    { DG_T0 pass(T_{}, F_{}, u0, 0); asm volatile("" ::"v"(accC)); DG_ADD(dg_chain) }
    { DG_T0 unit_barrier(min(u0 + kDenseRing - 1, u1 - 1) - (u0 + 1)); DG_ADD(dg_bar) }
    for (int u = u0 + 1; u < u1; ++u) {
      if (u + kDenseRing - 1 < u1) stage(u + kDenseRing - 1);
      { DG_T0 pass(T_{}, T_{}, u, u - 1); asm volatile("" ::"v"(accC)); DG_ADD(dg_chain) }
      { DG_T0 unit_barrier(min(u + kDenseRing - 1, u1 - 1) - (u + 1)); DG_ADD(dg_bar) }
      fold_columns(u - 1);
    }
    { DG_T0 pass(F_{}, T_{}, 0, u1 - 1); DG_ADD(dg_chain) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fold_columns(u1 - 1);
  }

#ifdef FM_DIAG_CLOCK
  dg_loop_end = __builtin_amdgcn_s_memtime();
#endif
#ifdef FM_DIAG_CLOCK       // the screening kernel's row-list slots of the padded rows (>= L, never used) carry 64 waves' stamps
  auto diag_stamp = [&]() {
  if (split == 0 && panel < 8 && (a.Lp - a.L) * a.slots >= 512 && lane < 8) {
    const float vals[8] = {(float)(__builtin_amdgcn_s_memtime() - dg0), (float)(__builtin_amdgcn_s_memrealtime() - dgr0),
                           (float)(u1 - u0), (float)dg_chain, 0.f, (float)dg_bar, (float)dg_pro,
                           (float)(__builtin_amdgcn_s_memtime() - dg_loop_end)};
    float vv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) vv = lane == q ? vals[q] : vv;
    a.diag[((long)b * a.Lp + a.L) * a.slots + (panel * 8 + wv) * 8 + lane] = vv;
  }
  };
  if constexpr (CONF) diag_stamp();
#endif
  if constexpr (CONF) return;
  // ---- the parked candidates -> the per-row / per-column slot lists: one entry per lane; the slot reservation (a
  // returning atomic: one memory round trip) is issued here and consumed behind the row reduction below ----
  int q_pos = -1, q_cpos = -1, q_key = 0;
  float q_x = 0.f;
  long q_row = 0, q_col = 0;
  {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nq = min(qn, kDenseQueue);
    if (lane < nq) {
      asm volatile("ds_read_b32 %0, %1" : "=v"(q_key) : "v"(qkey_a + lane * 4));
      asm volatile("ds_read_b32 %0, %1" : "=v"(q_x) : "v"(qx_a + lane * 4));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q_key), "+v"(q_x));
      q_row = (long)b * a.Lp + wrow0 + (q_key & 31);
      q_col = (long)b * a.Sp + (q_key >> 5);
      q_pos = atomicAdd(&a.cand_count[q_row], 1);
      q_cpos = atomicAdd(&a.ccand_count[q_col], 1);
    }
  }
  // ---- row sums of this workgroup's column range: reduce over the 32 lanes of each half ----
  if constexpr (SUMS) {
#pragma unroll
    for (int g = 0; g < 16; ++g) rstat[g] = half_sum32_d(rstat[g]);
    if (r == 0) {
      float* out = a.rowpart + ((long)b * a.splits + split) * a.Lp + wrow0 + 4 * h;
#pragma unroll
      for (int g = 0; g < 16; ++g) out[(g & 3) + 8 * (g >> 2)] = rstat[g];
    }
  }
  if (q_pos >= 0) {
    if (q_pos < a.slots) { a.cand_j[q_row * a.slots + q_pos] = q_key >> 5; a.cand_x[q_row * a.slots + q_pos] = q_x; }
    if (q_cpos < a.slots) { a.ccand_i[q_col * a.slots + q_cpos] = wrow0 + (q_key & 31); a.ccand_x[q_col * a.slots + q_cpos] = q_x; }
    if (q_pos >= a.slots || q_cpos >= a.slots) atomicOr(a.flags, RESCREEN ? (unsigned)FM_DEV_CANDIDATES : (unsigned)FM_INT_SCREEN_OVERFLOW);
  }
#ifdef FM_DIAG_CLOCK
  diag_stamp();
#endif
}

hipError_t launch_dense(const CoarseWs& w, char* base, float inv_ct, float thr, hipStream_t st, float* conf, int rescreen) {
  DenseArgs a;
  a.conf = conf;
  const bool offsets = conf != nullptr || rescreen;      // both read the log-softmax offsets of k_reduce_sums
  a.hi0 = (const _Float16*)(base + w.hi0); a.lo0 = (const _Float16*)(base + w.lo0);
  a.hi1 = (const _Float16*)(base + w.hi1); a.lo1 = (const _Float16*)(base + w.lo1);
  a.nmr = (const float*)(base + (offsets ? w.nmr2 : w.nmr)); a.nmc = (const float*)(base + (offsets ? w.nmc2 : w.nmc));
  a.rowpart = (float*)(base + w.rowB); a.colpart = (float*)(base + w.colB);
  a.dense_cnt = (const int*)(base + w.dense_cnt);
  a.dense_units = &((const Scalars*)(base + w.scalars))->dense_units;
  a.f16inv = (const float*)(base + w.f16inv);
  a.cand_count = (int*)(base + w.cand_count_b); a.cand_j = (int*)(base + w.cand_j_b); a.cand_x = (float*)(base + w.cand_x_b);
  a.ccand_count = (int*)(base + w.ccand_count_b); a.ccand_i = (int*)(base + w.ccand_i_b); a.ccand_x = (float*)(base + w.ccand_x_b);
  a.flags = (unsigned*)(base + w.scalars);
  a.diag = (float*)(base + w.cand_x);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.units = w.Sp / 32;
  // the row partials are [N][w.splits][Lp] (k_select / k_reduce_sums fold w.splits of them): the same number of splits
  // as the tile-based kernel it replaces, in units of 32 columns
  a.splits = w.splits;
  a.units_per_split = 2 * ((w.tiles + w.splits - 1) / w.splits);
  a.slots = w.slots;
  {
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.units_per_split * 32) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
    if (a.splits == 1) a.pgroup = w.panels;
  }
  a.k = inv_ct * kLog2e;
  a.lt = log2f(thr) - (rescreen ? 2e-4f : 0.f);      // the re-screening compares rounded log-softmax values: small guard
  const int blocks = w.N * a.splits * w.panels;
  hipError_t e = hipSuccess;
#define FM_DENSE_CASE(CC)                                                                        \
  case CC: {                                                                                     \
    static unsigned long long lds_set = 0, lds_set_c = 0, lds_set_r = 0;                         \
    if (conf) {      /* the samples the screening kernel served, then (exits at once without any) the dense kernel's */ \
      static unsigned long long lds_set_l = 0;                                                   \
      e = ensure_dynamic_lds(&k_dense<CC, kDenseConfLite>, kDenseRing * 2 * (CC / 16) * 1024, &lds_set_l); \
      if (e != hipSuccess) return e;                                                             \
      hipLaunchKernelGGL((k_dense<CC, kDenseConfLite>), dim3(blocks), dim3(512), kDenseRing * 2 * (CC / 16) * 1024, st, a);   \
      e = ensure_dynamic_lds(&k_dense<CC, kDenseConf>, kDenseRing * 2 * (CC / 16) * 1024, &lds_set_c); \
      if (e != hipSuccess) return e;                                                             \
      hipLaunchKernelGGL((k_dense<CC, kDenseConf>), dim3(blocks), dim3(512), kDenseRing * 2 * (CC / 16) * 1024, st, a);   \
    } else if (rescreen) {                                                                       \
      e = ensure_dynamic_lds(&k_dense<CC, kDenseRescreen>, kDenseRing * 2 * (CC / 16) * 1024, &lds_set_r); \
      if (e != hipSuccess) return e;                                                             \
      hipLaunchKernelGGL((k_dense<CC, kDenseRescreen>), dim3(blocks), dim3(512), kDenseRing * 2 * (CC / 16) * 1024, st, a);   \
    } else {                                                                                     \
      e = ensure_dynamic_lds(&k_dense<CC, kDenseSums>, kDenseRing * 2 * (CC / 16) * 1024, &lds_set);  \
      if (e != hipSuccess) return e;                                                             \
      hipLaunchKernelGGL((k_dense<CC, kDenseSums>), dim3(blocks), dim3(512), kDenseRing * 2 * (CC / 16) * 1024, st, a);  \
    }                                                                                            \
    break;                                                                                       \
  }
  switch (w.C) {
    FM_DENSE_CASE(64)
    FM_DENSE_CASE(128)
    FM_DENSE_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_DENSE_CASE
  return hipGetLastError();
}

}  // namespace fm
