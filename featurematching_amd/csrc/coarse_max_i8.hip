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
// Structure: one workgroup = 4 waves = a 256-row panel of image 0 x a range of 64-column tiles of image 1; each wave
// keeps its 64 rows (two 32-row blocks) as A fragments in 64 VGPRs for the whole sweep, so every B fragment read from
// LDS feeds two MFMAs on two independent accumulators; image-1 tiles (16 KiB at C = 256) stream through a 4-deep LDS
// ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB fragment block per instruction, refill pieces issued between MFMAs),
// handed over by counted vmcnt + raw s_barrier; B fragments are read ahead through inline-asm ds_read_b128 + counted
// lgkmcnt.  __launch_bounds__(256, 2): 213 VGPRs and 64 KiB of LDS leave room for two workgroups per CU.
#include <type_traits>

#include "fm_device.h"

namespace fm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MaxArgs {
  const signed char* q0; const signed char* q1;
  unsigned* rowmax_u; unsigned* colmax_u; float* umax;
  const float4* bstat0; const float4* bstat1; float* imgstat;   // block statistics of k_prep_split -> per-sample maxima
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
// Integer maxima as volatile asm: hipcc otherwise hoists the 32 running row maxima of a unit out of the MFMA gaps they
// are written into (identical code in both arms of the loop's last-tile branch) and runs them in one lump in front of
// the MFMA chain, where nothing hides them.
__device__ __forceinline__ void vmax_i(int& d, int s) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(d) : "v"(s)); }
__device__ __forceinline__ int vmax3_i(int x, int y, int z) {
  int d;
  asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}
__device__ __forceinline__ int halves_max_i(int v) {
  int p = v, q = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return max(p, q);
}

#ifndef FM_MAX_EPI
#define FM_MAX_EPI 3           // experiments: bit 0 = row maxima, bit 1 = column / unit maxima
#endif

template <int C>
__global__ __launch_bounds__(256, 2) void k_max_i8(MaxArgs a) {
  constexpr int KS8 = C / 32;                       // k-steps of 32 channels
  constexpr int TILE_BYTES = kTileCols * C;         // 64 columns x C bytes
  constexpr int NBUF = 4;
  constexpr int PIECES = 2 * KS8;                   // 1 KiB fragment blocks per tile: [column block 0/1][k-step]
  constexpr int PER_WAVE = PIECES / 4;              // C >= 64: every wave brings >= 1 block of a tile
  constexpr int PF = KS8 < 4 ? KS8 : 4;             // B-fragment read-ahead
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per tile (mod 3): q_encode'd column maxima of 64 columns (ds_max_u32); words 192..197: the sample's statistics
  __shared__ unsigned s_colmax[3 * 64 + 8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef FM_DIAG_CLOCK       // diagnostic build only: shader-clock stamps per phase of every wave (tools/diag_max.py)
  const unsigned long long dg0 = __builtin_amdgcn_s_memtime();
  const unsigned long long dgr0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz clock
  unsigned long long dg_pro = 0, dg_mfma = 0, dg_epi = 0, dg_bar = 0;
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

  const int wrow0 = panel * kPanelRows + wv * 64;   // this wave: rows wrow0 .. wrow0 + 63, two 32-row blocks
  const signed char* plane1 = a.q1 + (long)b * a.Sp * C;
  auto stage_piece = [&](int t, int buf, int n) {                   // this wave's n-th 1 KiB block of tile t
    const int blk = wv * PER_WAVE + n;                              // (cb, ks)
    const signed char* src = plane1 + ((long)(2 * t) * KS8 + blk) * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + buf * TILE_BYTES + blk * 1024),
                                     16, 0, 0);
  };
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int n = 0; n < PER_WAVE; ++n) stage_piece(t, buf, n);
  };
  // The first workgroup of every sample also folds the block statistics of k_prep_split (largest L1 norm, largest
  // clipped mass, largest |x| of the 32-row blocks of both images) into the sample's six maxima: the screening kernel's
  // every wave needs them for its margins, and used to reduce the ~300 block records itself (16 loads and six wave
  // reductions per wave).  The loads go out ahead of the tile prefetch (older in the vmcnt order), the fold runs
  // behind the first barrier.
  const bool stat_wg = panel == 0 && split == 0;         // (uniform)
  float st6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (stat_wg) {
    const int nb0 = a.Lp / 32;
    for (int i = tid; i < nb0; i += 256) {
      const float4 v = a.bstat0[(long)b * nb0 + i];
      st6[0] = fmaxf(st6[0], v.x); st6[1] = fmaxf(st6[1], v.y); st6[2] = fmaxf(st6[2], v.z);
    }
    for (int i = tid; i < nunits; i += 256) {
      const float4 v = a.bstat1[(long)b * nunits + i];
      st6[3] = fmaxf(st6[3], v.x); st6[4] = fmaxf(st6[4], v.y); st6[5] = fmaxf(st6[5], v.z);
    }
  }
#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d)
    if (t0 + d < t1) stage(t0 + d, d);

  // this wave's 64 rows as A fragments: one contiguous 1 KiB block per row block and k-step of the fragment-major plane
  v4i aq0[KS8], aq1[KS8];
  {
    const signed char* src = a.q0 + (((long)b * a.Lp + wrow0) / 32 * KS8 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) {
      aq0[ks] = *reinterpret_cast<const v4i*>(src + ks * 1024);
      aq1[ks] = *reinterpret_cast<const v4i*>(src + (KS8 + ks) * 1024);
    }
  }
  const unsigned colmax_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)s_colmax;
  if (tid < 200) asm volatile("ds_write_b32 %0, %1" ::"v"(colmax_a + tid * 4), "v"(0u) : "memory");

  int rstat0[16], rstat1[16];    // running maxima of q_i . q_j over the columns this lane has seen (row block 0 / 1)
#pragma unroll
  for (int g = 0; g < 16; ++g) rstat0[g] = rstat1[g] = kQMasked;
  // padded rows (>= L: zero descriptors) must not win a column or unit maximum
  const bool row_edge0 = wrow0 + 32 > a.L, row_edge1 = wrow0 + 64 > a.L;       // wave-uniform
  // (a wave with nothing but padding rows - the last 64 rows of a 640x480 pair's last panel - would run the masks'
  // 32 compares + selects per unit for nothing, and its workgroup's other waves would wait for it at every barrier)
  const bool dead = wrow0 >= a.L;

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
  for (int ks = 0; ks < KS8; ++ks) { asm volatile("" ::"v"(aq0[ks])); asm volatile("" ::"v"(aq1[ks])); }
  if (stat_wg) {
    // non-negative floats (+inf = a block with a bad value) order like their bit patterns: ds_max_u32 (the words were
    // cleared in front of the barrier above); read back and stored behind the sweep's barriers, at the end
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float v = st6[q];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
      if (lane == 0) atomicMax(&s_colmax[192 + q], __float_as_uint(v));
    }
  }
#ifdef FM_DIAG_CLOCK
  dg_pro = __builtin_amdgcn_s_memtime() - dg0;
#endif

  // every wave's column maxima of tile t are in LDS once a barrier separates this from their epilogues: the wave whose
  // turn it is publishes them and clears the words for tile t + 3
  auto fold_columns = [&](int t) {
    if (wv != (t & 3)) return;
    const unsigned ad = colmax_a + (((t - t0) % 3) * 64 + lane) * 4;
    unsigned cv;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(ad) : "memory");
    asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(0u) : "memory");
    if (t * kTileCols + lane < a.S && cv != 0u)
      __hip_atomic_fetch_max(a.colmax_u + (long)b * a.Sp + t * kTileCols + lane, cv, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  };

  // Software pipeline over the units, written out instruction slot by instruction slot.  A unit is 64 rows x 32
  // columns: 2 x KS8 MFMAs on two independent accumulators (row block 0 / 1), every B fragment read from LDS ONCE
  // for both - with 32 rows per wave the four SIMDs' ds_read_b128 (1 KiB per 32-cycle MFMA each) need all of the
  // LDS' 128 B/clk, which capped the matrix cores at ~half their rate.  Behind each MFMA of the NEXT unit
  // (accumulators an0 / an1) sits one slice of the epilogue of the CURRENT unit (ac0 / ac1: rows in registers,
  // columns on lanes; integers only): a wave issues in order, so the integer instructions only run under the matrix
  // core's 32 cycles per MFMA if they sit between the MFMAs in program order (sched_barrier keeps hipcc from
  // regrouping them).  Slices: padding masks (2) | row maxima (4) | column maximum trees (2) | halves (2) | LDS column
  // maximum | unit maxima (one DPP chain for both row blocks) | their store.
  auto pipe = [&](auto do_next, auto do_cur, int un, v16i& an0, v16i& an1, int uc, v16i& ac0, v16i& ac1, int ts = -1, int tsbuf = 0) {
    constexpr bool DN = decltype(do_next)::value, DC = decltype(do_cur)::value;
    if (dead) {                        // nothing but padding rows: this wave only helps to stage the tiles
      if (DN && DC && ts >= 0) stage(ts, tsbuf);
      return;
    }
    const unsigned base = lds0 + (((un >> 1) - t0) % NBUF) * TILE_BYTES + (un & 1) * (KS8 * 1024) + lane * 16;
    constexpr int RING = PF + 1;
    v4i bq[RING];
    auto issue_ = [](auto ksc, v4i (&bq_)[RING], unsigned base_) {    // (the k-step's offset rides in the instruction)
      constexpr int ks = decltype(ksc)::value;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq_[ks % RING]) : "v"(base_), "n"(ks * 1024));
    };
#define issue(...) issue_(__VA_ARGS__, bq, base)
    auto mask_edge = [&](v16i& ac, bool row_edge, int row00) {
      const int ucol0 = uc * 32;
      if (row_edge || ucol0 + 32 > a.S) {              // padded rows (>= L) / columns (>= S) never count
        const bool cok = ucol0 + r < a.S;
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (!cok || row00 + (g & 3) + 8 * (g >> 2) + 4 * h >= a.L) ac[g] = kQMasked;
      }
    };
    // The epilogue of a unit as 16 instruction groups, one behind each MFMA of a C = 256 unit (C = 128 / 64: two / four
    // groups per MFMA).  R = one running row maximum (32 per unit), T0 / T1 = one v_max3 of the column-maximum tree of
    // row block 0 / 1 (8 each):
    //    0..3  : R R T0            4..7 : R R T0 T1          8 : R R T1 + halves of block 0       9..11 : R R T1
    //   12     : R R + halves of block 1          13 : R R + LDS column maximum + first DPP step of the unit maxima
    //   14     : R R + DPP step    15 : R R + last three DPP steps + store
    // <= 6 vector instructions per group: what one 32-cycle MFMA hides (MI355X_MICROARCH.md, issue costs).  All of them
    // are volatile asm, which keeps hipcc from regrouping them (it hoisted the row maxima out of the gaps as common code
    // of the loop's two arms); the two wait states a DPP operand needs after the write of its register are other
    // groups' instructions or s_nop 1.
    int ta[8], tb[8], c0 = 0, c1 = 0, um = 0;
    auto R = [&](int k) {
      if (!(FM_MAX_EPI & 1)) { if (k == 0) { vmax_i(rstat0[0], ac0[0]); vmax_i(rstat1[0], ac1[0]); } return; }
      if (k < 16) vmax_i(rstat0[k], ac0[k]); else vmax_i(rstat1[k - 16], ac1[k - 16]);
    };
    auto T = [&](const v16i& ac, int (&t)[8], int i) {
      if (!(FM_MAX_EPI & 2)) { if (i == 0) t[7] = ac[15]; return; }
      if (i < 5) t[i] = vmax3_i(ac[3 * i], ac[3 * i + 1], ac[3 * i + 2]);
      else if (i == 5) t[5] = vmax3_i(t[0], t[1], t[2]);
      else if (i == 6) t[6] = vmax3_i(t[3], t[4], ac[15]);
      else { t[7] = t[5]; vmax_i(t[7], t[6]); }
    };
    auto dpp = [&](auto step) {
      constexpr int st = decltype(step)::value;
      if (st == 0) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(um));
      else if (st == 1) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(um));
      else if (st == 2) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(um));
      else if (st == 3) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(um));
      else asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(um));
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    auto group = [&](int s) {
      if (s == 0) { mask_edge(ac0, row_edge0, wrow0); mask_edge(ac1, row_edge1, wrow0 + 32); }
      R(2 * s); R(2 * s + 1);
      if (s < 8) T(ac0, ta, s);
      if (s >= 4 && s < 12) T(ac1, tb, s - 4);
      if (!(FM_MAX_EPI & 2)) { if (s == 15) { vmax_i(rstat0[1], ta[7]); vmax_i(rstat1[1], tb[7]); } return; }
      if (s == 8) c0 = halves_max_i(ta[7]);                           // this lane's column over row block 0
      if (s == 12) c1 = halves_max_i(tb[7]);                          // ... over row block 1
      if (s == 13) {
        // the 4 waves' maxima of a column meet in LDS (both halves of the wave hold the same value: no lane mask)
        asm volatile("ds_max_u32 %0, %1" ::"v"(colmax_a + ((((uc >> 1) - t0) % 3) * 64 + (uc & 1) * 32 + r) * 4),
                     "v"(q_encode(max(c0, c1))) : "memory");
        um = h ? c1 : c0;                                             // unit maxima: block 0 in lanes 0..31, block 1 in 32..63
        asm volatile("s_nop 1" : "+v"(um));
        dpp(I0{});
      }
      if (s == 14) dpp(I1{});
      if (s == 15) {
        dpp(I2{}); dpp(I3{}); dpp(I4{});
        if (r == 31) a.umax[((long)b * (a.Lp / 32) + wrow0 / 32 + h) * nunits + uc] = (float)um;
      }
    };
    constexpr int NSLOT = 2 * KS8;
    auto slot = [&](int sidx) {                         // the groups of MFMA slot sidx (of NSLOT per unit)
#pragma unroll
      for (int g = sidx * 16 / NSLOT; g < (sidx + 1) * 16 / NSLOT; ++g) group(g);
    };
    if (DN) {
      issue(I0{});
      if constexpr (PF > 1) issue(I1{});
      if constexpr (PF > 2) issue(I2{});
      if constexpr (PF > 3) issue(I3{});
    } else {
      // (asm operands are invisible to hipcc's hazard recogniser: the last MFMA's passes must have drained)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    // k-steps as a compile-time recursion (the LDS offsets and wait counts are instruction immediates)
    auto kstep = [&](auto self, auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if (DN) {
        if constexpr (ks + PF < KS8) issue(std::integral_constant<int, ks + PF>{});
        constexpr int ahead = (KS8 - 1 - ks) < PF ? (KS8 - 1 - ks) : PF;     // k-steps issued beyond ks
        static_assert(PF <= 4, "lgkmcnt immediates below cover at most 4 reads in flight");
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[ks % RING]) : "n"(ahead));
        if (ks == 0) {
          v16i z;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = 0;
          an0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq0[ks], bq[ks % RING], z, 0, 0, 0);
        } else {
          an0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq0[ks], bq[ks % RING], an0, 0, 0, 0);
        }
      }
      if (DC) slot(2 * ks);
      if (DN && DC) __builtin_amdgcn_sched_barrier(0);
      if (DN) {
        if (ks == 0) {
          v16i z;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = 0;
          an1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq1[ks], bq[ks % RING], z, 0, 0, 0);
        } else {
          an1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq1[ks], bq[ks % RING], an1, 0, 0, 0);
        }
      }
      if (DC) slot(2 * ks + 1);
      // the ring refill rides in the gaps of this pipe (an LDS-DMA instruction issued between MFMAs costs a fraction of
      // one issued in a burst in front of them): piece n behind k-step (n + 1) KS8 / PER_WAVE - 1
#ifndef FM_ABL_MAX_NODMA    // (timing-only ablation build: the ring is not refilled - wrong maxima; profiles/r06_ab_max_ablations.txt)
      if (DN && DC && ((ks + 1) * PER_WAVE) % KS8 == 0 && ts >= 0) stage_piece(ts, tsbuf, (ks + 1) * PER_WAVE / KS8 - 1);
#endif
      if (DN && DC) __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + 1 < KS8) self(self, std::integral_constant<int, ks + 1>{});
    };
    kstep(kstep, I0{});
#undef issue
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  // Per tile t (units 2t in accA*, 2t+1 in accB*):
  //   refill the ring slot of tile t-1 | MFMAs(2t+1) -> B with the epilogue of 2t (A) between them |
  //   barrier: every wave has read tile t, tile t+1 has landed | publish tile t-1's column maxima |
  //   MFMAs(2t+2) -> A with the epilogue of 2t+1 (B) between them
  v16i accA0, accA1, accB0, accB1;
  if (t0 < t1) {
    pipe(T_{}, F_{}, 2 * t0, accA0, accA1, 0, accB0, accB1);
    // (the last tile is peeled: a branch on "is there a next tile" inside the loop makes hipcc carry the 32 running
    // row maxima and the accumulators through register copies - 64 v_mov per tile)
    for (int t = t0; t + 1 < t1; ++t) {
      { DG_T0 pipe(T_{}, T_{}, 2 * t + 1, accB0, accB1, 2 * t, accA0, accA1,
                   t + NBUF - 1 < t1 ? t + NBUF - 1 : -1, (t - t0 + NBUF - 1) % NBUF); DG_ADD(dg_mfma) }
      { DG_T0 tile_barrier(min(t + NBUF - 1, t1 - 1) - (t + 1)); DG_ADD(dg_bar) }
      if (t > t0) fold_columns(t - 1);
      { DG_T0 pipe(T_{}, T_{}, 2 * t + 2, accA0, accA1, 2 * t + 1, accB0, accB1); DG_ADD(dg_epi) }
    }
    const int t = t1 - 1;
    { DG_T0 pipe(T_{}, T_{}, 2 * t + 1, accB0, accB1, 2 * t, accA0, accA1); DG_ADD(dg_mfma) }
    { DG_T0 tile_barrier(0); DG_ADD(dg_bar) }
    if (t > t0) fold_columns(t - 1);
    { DG_T0 pipe(F_{}, T_{}, 0, accA0, accA1, 2 * t + 1, accB0, accB1); DG_ADD(dg_epi) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fold_columns(t1 - 1);
  }
  if (stat_wg && tid < 6) a.imgstat[(long)b * 8 + tid] = __uint_as_float(s_colmax[192 + tid]);

#ifdef FM_DIAG_CLOCK
  const unsigned long long dg_tail0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- row maxima of this workgroup's column range ----
  // 32 registers (2 row blocks x 16 rows per lane half) x 32 lanes (columns) -> one row maximum per lane by a
  // transposing reduction: at every level a lane keeps half of its registers and hands the other half to its partner,
  // 32 + 16 + 8 + 4 + 2 = 62 maxima instead of 32 x 5 DPP steps, and ONE atomic instruction for the wave's 64 rows.
  //   level E (lane ^ 16): v_permlane16_swap exchanges the odd rows of X with the even rows of Y, max(X, Y) then holds
  //                        X's maximum in rows 0 / 2 and Y's in rows 1 / 3
  //   level A (15 - i)   : row_mirror, banks 0-1 keep X, banks 2-3 take Y's maximum (bank-masked writes into X)
  //   level B (7 - i)    : row_half_mirror, banks 0 / 2 keep X, banks 1 / 3 take Y's
  //   level C (3 - i), D (i ^ 1): inside a quad, by select + quad_perm
  // lane bits b4..b0 (of lane & 31) -> register b4 + 2 b3 + 4 b2 + 8 b1 + 16 b0.
  if (!dead) {
    int V[32];
#pragma unroll
    for (int g = 0; g < 16; ++g) { V[g] = rstat0[g]; V[16 + g] = rstat1[g]; }
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int m = 0; m < 16; ++m) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(V[2 * m]), "+v"(V[2 * m + 1]));
    int W[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) W[m] = max(V[2 * m], V[2 * m + 1]);
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int m = 0; m < 8; ++m)
      asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(W[2 * m]));
#pragma unroll
    for (int m = 0; m < 8; ++m)
      asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(W[2 * m]) : "v"(W[2 * m + 1]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m)
      asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(W[4 * m]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m)
      asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(W[4 * m]) : "v"(W[4 * m + 2]));
    asm volatile("s_nop 1" ::: "memory");
    const bool b1 = (lane >> 1) & 1, b0 = lane & 1;
    int S[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int X = W[8 * m], Y = W[8 * m + 4];
      const int own = b1 ? Y : X, other = b1 ? X : Y;
      S[m] = max(own, __builtin_amdgcn_update_dpp(other, other, 0x1B, 0xf, 0xf, false));       // lane 3 - i of the quad
    }
    const int own = b0 ? S[1] : S[0], other = b0 ? S[0] : S[1];
    const int Q = max(own, __builtin_amdgcn_update_dpp(other, other, 0xB1, 0xf, 0xf, false));    // lane i ^ 1
    const int vi = ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1) + 8 * (int)b1 + 16 * (int)b0;
    const int g = vi & 15;
    const int row = wrow0 + (vi >> 4) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
    if (row < a.L && Q > kQMasked)
      __hip_atomic_fetch_max(a.rowmax_u + (long)b * a.Lp + row, q_encode(Q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef FM_DIAG_CLOCK
  if (lane < 8) {
    const float vals[8] = {(float)(__builtin_amdgcn_s_memtime() - dg0), (float)dg_pro, (float)dg_mfma, (float)dg_epi,
                           (float)dg_bar, (float)(__builtin_amdgcn_s_memrealtime() - dgr0), (float)(dgr0 & 0xffffff), (float)(__builtin_amdgcn_s_memtime() - dg_tail0)};
    float vv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) vv = lane == q ? vals[q] : vv;
    a.diag[((long)blockIdx.x * 4 + wv) * 8 + lane] = vv;
  }
#endif
}

hipError_t launch_max_i8(const CoarseWs& w, char* base, hipStream_t st) {
  MaxArgs a;
  a.q0 = (const signed char*)(base + w.q0); a.q1 = (const signed char*)(base + w.q1);
  a.rowmax_u = (unsigned*)(base + w.rowmax_u); a.colmax_u = (unsigned*)(base + w.colmax_u);
  a.umax = (float*)(base + w.umax);
  a.bstat0 = (const float4*)(base + w.bstat0); a.bstat1 = (const float4*)(base + w.bstat1);
  a.imgstat = (float*)(base + w.imgstat);
  a.diag = (float*)(base + w.rowB);      // (diagnostic builds run on a full-size workspace)
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.tiles = w.tiles;
  a.tiles_per_split = (w.tiles + w.splits0 - 1) / w.splits0;
  a.splits = (w.tiles + a.tiles_per_split - 1) / a.tiles_per_split;      // (no empty split)
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
    hipLaunchKernelGGL(k_max_i8<CC>, dim3(blocks), dim3(256), 4 * kTileCols * CC, st, a);      \
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
