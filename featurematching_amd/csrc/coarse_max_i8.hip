// Coarse stage, the max pass on the int8 screening planes.
//
// First sweep of network/utils/coarse_matching_new.py:64-68 (all-pairs correlation): it only has to tell the
// sum kernels WHERE the mass of the dual softmax is - per-row and per-column maxima (stabilisers) and the
// maximum of every 32 x 32 unit (which units are alive) - so it runs on the int8 planes of k_prep_split:
// v_mfma_i32_32x32x32_i8 does twice the k per instruction of the f16 form, the planes have half the bytes
// (prologue, LDS-DMA, LDS reads all halve), and the quantisation error is bounded rigorously (fm_device.h).
//   screening product  x~_ij = sigma_0 * sigma_1 * (q_i . q_j)   (exact integer dot; ONE step per image)
// Because the step is uniform per image, x~ is ordered like the integer dot product: the whole epilogue is integer
// arithmetic on the accumulators - one v_max_i32 per register for the row maxima, a v_max3_i32 tree for the column
// maxima, six DPP steps for the unit maximum - about 35 vector instructions per 32 x 32 unit against the 8 MFMAs
// (256 cycles) that produce it.  (With one step per descriptor, as in round 2, every element needed convert + two
// scalings + two float maxima: 130 instructions per unit, 2.5x the matrix cores' time.)
// Maxima are published as biased integer codes with atomicMax (exact, order independent): no partial arrays and no
// reduction kernel.
//
// Structure (as the dense sum kernel k_corr): one workgroup = 8 waves = a 256-row panel of image 0 x a range of
// 64-column tiles of image 1; each wave keeps its 32 rows as A fragments in 32 VGPRs for the whole sweep; image-1
// tiles (16 KiB at C = 256) stream through a 4-deep LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB fragment
// block per instruction), handed over by counted vmcnt + raw s_barrier; B fragments are read ahead through
// inline-asm ds_read_b128 + counted lgkmcnt.  The LDS footprint (64 KiB) and < 100 VGPRs leave room for two
// workgroups per CU.
#include <type_traits>

#include "fm_device.h"

namespace fm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MaxArgs {
  const signed char* q0; const signed char* q1;
  unsigned* rowmax_u; unsigned* colmax_u; float* umax;
  float* diag;            // diagnostic build: stamp buffer
  int L, S, Lp, Sp, panels, tiles, splits, tiles_per_split, pgroup;
};

__device__ __forceinline__ int xcd_remap_m(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}
// max over the 32 lanes that share lane>>5; valid in lanes 16..31 / 48..63.  One v_max_i32 with a DPP operand per
// step (hipcc keeps v_mov_b32_dpp + v_max apart when written with the update_dpp builtin): xor 1, xor 2, half mirror
// and mirror reduce every row of 16 lanes, row_bcast:15 hands row 0's result to row 1 (and row 2's to row 3).
// s_nop 1: the two wait states between a VALU write and a DPP read of the same register.
__device__ __forceinline__ int half_max32_hi_i(int v) {
  asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
               : "+v"(v));
  return v;
}
__device__ __forceinline__ int halves_max_i(int v) {
  int p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return max(p, q);
}

template <int C>
__global__ __launch_bounds__(512) void k_max_i8(MaxArgs a) {
  constexpr int KS8 = C / 32;                       // k-steps of 32 channels
  constexpr int TILE_BYTES = kTileCols * C;         // 64 columns x C bytes
  constexpr int NBUF = 4;
  constexpr int PIECES = 2 * KS8;                   // 1 KiB fragment blocks per tile: [column block 0/1][k-step]
  constexpr int PER_WAVE = PIECES >= 8 ? PIECES / 8 : 1;   // (C = 64: the 8 waves bring the 4 blocks twice - harmless)
  constexpr int PF = KS8 < 4 ? KS8 : 4;             // B-fragment read-ahead
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned s_colmax[3 * 64];             // per tile (mod 3): q_encode'd column maxima of 64 columns (ds_max_u32)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps per phase of every wave (tools/diag_max.py)
  const unsigned long long dg0 = __builtin_amdgcn_s_memtime();
  unsigned long long dg_pro = 0, dg_mfma = 0, dg_epi = 0, dg_bar = 0, dg_stage = 0;
#define DG_T0 const unsigned long long dg_t = __builtin_amdgcn_s_memtime();
#define DG_ADD(x) x += __builtin_amdgcn_s_memtime() - dg_t;
#else
#define DG_T0
#define DG_ADD(x)
#endif

  // workgroup order: sample, groups of a.pgroup panels, split-major inside a group, through the bijective XCD
  // remap - one XCD's share is a compact (panels x splits) block (speed only)
  int kk = xcd_remap_m(blockIdx.x, gridDim.x);
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
  const int nunits = a.Sp / 32;

  const int wrow0 = panel * kPanelRows + wv * 32;
  const signed char* plane1 = a.q1 + (long)b * a.Sp * C;
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int n = 0; n < PER_WAVE; ++n) {
      const int blk = (wv * PER_WAVE + n) % PIECES;                 // (cb, ks)
      const signed char* src = plane1 + ((long)(2 * t) * KS8 + blk) * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + buf * TILE_BYTES + blk * 1024),
                                       16, 0, 0);
    }
  };
#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d)
    if (t0 + d < t1) stage(t0 + d, d);

  // this wave's 32 rows as A fragments: one contiguous 1 KiB block per k-step of the fragment-major plane
  v4i aq[KS8];
  {
    const signed char* src = a.q0 + (((long)b * a.Lp + wrow0) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) aq[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
  }
  const unsigned colmax_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)s_colmax;
  if (tid < 192) asm volatile("ds_write_b32 %0, %1" ::"v"(colmax_a + tid * 4), "v"(0u) : "memory");

  int rstat[16];                 // running maxima of q_i . q_j over the columns this lane has seen
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat[g] = kQMasked;
  // padded rows (>= L: zero descriptors) must not win a column or unit maximum
  const bool row_edge = wrow0 + 32 > a.L;       // wave-uniform

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  auto tile_barrier = [&](int tiles_after) {
    if (tiles_after <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (tiles_after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(2 * PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  tile_barrier(min(NBUF - 2, t1 - t0 - 1));     // first tile landed, column maxima cleared
  // name the registers loaded before the loop: hipcc then waits for them here and not (with vmcnt(0), i.e. also
  // for the tile prefetch) at their first use inside the loop
#pragma unroll
  for (int ks = 0; ks < KS8; ++ks) asm volatile("" ::"v"(aq[ks]));
#ifdef FM_DIAG_CLOCK
  dg_pro = __builtin_amdgcn_s_memtime() - dg0;
#endif

  // every wave's column maxima of tile t are in LDS once a barrier separates this from their epilogues: the wave whose
  // turn it is publishes them and clears the words for tile t + 3
  auto fold_columns = [&](int t) {
    if (wv != (t & 7)) return;
    const unsigned ad = colmax_a + (((t - t0) % 3) * 64 + lane) * 4;
    unsigned cv;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(ad) : "memory");
    asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(0u) : "memory");
    if (t * kTileCols + lane < a.S && cv != 0u)
      __hip_atomic_fetch_max(a.colmax_u + (long)b * a.Sp + t * kTileCols + lane, cv, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  };

  // Software pipeline over the units, written out instruction slot by instruction slot: the 8 MFMAs of the NEXT unit
  // (accumulator `an`) with one slice of the epilogue of the CURRENT unit (accumulator `ac`: rows in registers, columns
  // on lanes; integers only) behind each of them - a wave issues in order, so the ~45 integer instructions of an
  // epilogue only run under the matrix core's 32 cycles per MFMA if they sit between the MFMAs in program order
  // (sched_barrier keeps hipcc from regrouping them).  Slices: padding masks | row maxima (2) | column maximum tree |
  // halves | LDS column maximum | unit maximum (DPP) | its store.
  auto pipe = [&](auto do_next, auto do_cur, int un, v16i& an, int uc, v16i& ac) {
    constexpr bool DN = decltype(do_next)::value, DC = decltype(do_cur)::value;
    const unsigned base = lds0 + (((un >> 1) - t0) % NBUF) * TILE_BYTES + (un & 1) * (KS8 * 1024) + lane * 16;
    constexpr int RING = PF + 1;
    v4i bq[RING];
    auto issue = [&](int ks) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(bq[ks % RING]) : "v"(base + (unsigned)(ks * 1024)));
    };
    int cstat = 0, um = 0;
    auto slice = [&](int sl) {
      if (sl == 0) {
        const int ucol0 = uc * 32;
        if (row_edge || ucol0 + 32 > a.S) {              // padded rows (>= L) / columns (>= S) never count
          const bool cok = ucol0 + r < a.S;
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (!cok || wrow0 + (g & 3) + 8 * (g >> 2) + 4 * h >= a.L) ac[g] = kQMasked;
        }
      } else if (sl == 1) {
#pragma unroll
        for (int g = 0; g < 8; ++g) rstat[g] = max(rstat[g], ac[g]);
      } else if (sl == 2) {
#pragma unroll
        for (int g = 8; g < 16; ++g) rstat[g] = max(rstat[g], ac[g]);
      } else if (sl == 3) {
        const int c01 = max(max(ac[0], ac[1]), ac[2]), c23 = max(max(ac[3], ac[4]), ac[5]);
        const int c45 = max(max(ac[6], ac[7]), ac[8]), c67 = max(max(ac[9], ac[10]), ac[11]);
        const int c89 = max(max(ac[12], ac[13]), ac[14]);
        cstat = max(max(max(c01, c23), c45), max(max(c67, c89), ac[15]));
      } else if (sl == 4) {
        cstat = halves_max_i(cstat);                                 // this lane's column over the wave's 32 rows
      } else if (sl == 5) {
        if (h == 0)                                                   // the 8 waves' maxima of a column meet in LDS
          asm volatile("ds_max_u32 %0, %1" ::"v"(colmax_a + ((((uc >> 1) - t0) % 3) * 64 + (uc & 1) * 32 + r) * 4),
                       "v"(q_encode(cstat)) : "memory");
      } else if (sl == 6) {
        um = half_max32_hi_i(cstat);                                 // unit maximum (lanes 16..31, 48..63)
      } else {
        if (lane == 63) a.umax[((long)b * (a.Lp / 32) + wrow0 / 32) * nunits + uc] = (float)um;
      }
    };
    if (DN) {
#pragma unroll
      for (int ks = 0; ks < PF && ks < KS8; ++ks) issue(ks);
    }
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) {
      if (DN) {
        if (ks + PF < KS8) issue(ks + PF);
        const int ahead = (KS8 - 1 - ks) < PF ? (KS8 - 1 - ks) : PF;     // k-steps issued beyond ks
        if (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[ks % RING]));
        else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq[ks % RING]));
        else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[ks % RING]));
        else if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[ks % RING]));
        else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[ks % RING]));
        static_assert(PF <= 4, "lgkmcnt ladder above covers at most 4 reads in flight");
        if (ks == 0) {
          v16i z;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = 0;
          an = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[ks], bq[ks % RING], z, 0, 0, 0);
        } else {
          an = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[ks], bq[ks % RING], an, 0, 0, 0);
        }
      }
      if (DC) {        // the epilogue slices of this k-step (8 slices over KS8 steps)
#pragma unroll
        for (int sl = ks * 8 / KS8; sl < (ks + 1) * 8 / KS8; ++sl) slice(sl);
      }
      if (DN && DC) __builtin_amdgcn_sched_barrier(0);
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  // Per tile t (units 2t in accA, 2t+1 in accB):
  //   refill the ring slot of tile t-1 | MFMAs(2t+1) -> B with the epilogue of 2t (A) between them |
  //   barrier: every wave has read tile t, tile t+1 has landed | publish tile t-1's column maxima |
  //   MFMAs(2t+2) -> A with the epilogue of 2t+1 (B) between them
  v16i accA, accB;
  if (t0 < t1) pipe(T_{}, F_{}, 2 * t0, accA, 0, accB);
  for (int t = t0; t < t1; ++t) {
    if (t + NBUF - 1 < t1) { DG_T0 stage(t + NBUF - 1, (t - t0 + NBUF - 1) % NBUF); DG_ADD(dg_stage) }
    { DG_T0 pipe(T_{}, T_{}, 2 * t + 1, accB, 2 * t, accA); DG_ADD(dg_mfma) }
    { DG_T0 tile_barrier(min(t + NBUF - 1, t1 - 1) - (t + 1)); DG_ADD(dg_bar) }
    if (t > t0) fold_columns(t - 1);
    { DG_T0
      if (t + 1 < t1) pipe(T_{}, T_{}, 2 * t + 2, accA, 2 * t + 1, accB);
      else pipe(F_{}, T_{}, 0, accA, 2 * t + 1, accB);
      DG_ADD(dg_epi) }
  }
  if (t0 < t1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fold_columns(t1 - 1);
  }

#ifdef FM_DIAG_CLOCK
  const unsigned long long dg_tail0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- row maxima of this workgroup's column range ----
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat[g] = half_max32_hi_i(rstat[g]);
  if (r == 31) {
    unsigned* out = a.rowmax_u + (long)b * a.Lp + wrow0 + 4 * h;
#pragma unroll
    for (int g = 0; g < 16; ++g)
      if (wrow0 + 4 * h + (g & 3) + 8 * (g >> 2) < a.L && rstat[g] > kQMasked)
        __hip_atomic_fetch_max(out + (g & 3) + 8 * (g >> 2), q_encode(rstat[g]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef FM_DIAG_CLOCK
  if (lane < 8) {
    const float vals[8] = {(float)(__builtin_amdgcn_s_memtime() - dg0), (float)dg_pro, (float)dg_mfma, (float)dg_epi,
                           (float)dg_bar, (float)dg_stage, (float)(2 * (t1 - t0)), (float)(__builtin_amdgcn_s_memtime() - dg_tail0)};
    float vv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) vv = lane == q ? vals[q] : vv;
    a.diag[((long)blockIdx.x * 8 + wv) * 8 + lane] = vv;
  }
#endif
}

hipError_t launch_max_i8(const CoarseWs& w, char* base, hipStream_t st) {
  MaxArgs a;
  a.q0 = (const signed char*)(base + w.q0); a.q1 = (const signed char*)(base + w.q1);
  a.rowmax_u = (unsigned*)(base + w.rowmax_u); a.colmax_u = (unsigned*)(base + w.colmax_u);
  a.umax = (float*)(base + w.umax);
  a.diag = (float*)(base + w.rowB);      // (diagnostic builds run on a full-size workspace)
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.tiles = w.tiles;
  a.splits = w.splits0; a.tiles_per_split = (w.tiles + a.splits - 1) / a.splits;
  {
    // one XCD runs ~blocks/8 workgroups: make its block of (panels x splits) as square as the bytes are
    const int blocks_all = w.N * a.splits * w.panels;
    const float share = fmaxf(1.f, (float)blocks_all / 8.f);
    int pgr = (int)lroundf(sqrtf(share * (float)(a.tiles_per_split * kTileCols) / (float)kPanelRows));
    a.pgroup = pgr < 1 ? 1 : (pgr > w.panels ? w.panels : pgr);
    if (a.splits == 1) a.pgroup = w.panels;
  }
  const int blocks = w.N * a.splits * w.panels;
  hipError_t e = hipSuccess;
#define FM_MAX_CASE(CC)                                                                        \
  case CC: {                                                                                   \
    static unsigned long long lds_set = 0;                                                     \
    e = ensure_dynamic_lds(&k_max_i8<CC>, 4 * kTileCols * CC, &lds_set);                       \
    if (e != hipSuccess) return e;                                                             \
    hipLaunchKernelGGL(k_max_i8<CC>, dim3(blocks), dim3(512), 4 * kTileCols * CC, st, a);      \
    break;                                                                                     \
  }
  switch (w.C) {
    FM_MAX_CASE(64)
    FM_MAX_CASE(128)
    FM_MAX_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_MAX_CASE
  return hipGetLastError();
}

}  // namespace fm
