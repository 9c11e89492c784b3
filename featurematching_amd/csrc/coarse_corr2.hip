// Coarse stage, the correlation sweeps - second-generation structure (selected with -DFM_CORR_V2).
//
// Same mathematics and outputs as coarse_corr.hip (see its header: pass A = maxima from the f16 "hi"
// product, pass B = float32-equivalent 3-product sums + candidate screening, pass C = conditional exact
// screening); only the mapping onto the CU differs:
//
//   * one workgroup = 4 waves = ONE WAVE PER SIMD, each wave owns 64 rows (two 32-row MFMA tiles) of the
//     256-row panel and the whole 512-entry register file of its SIMD: 256 registers hold its A
//     fragments (hi + lo, all of K), 64 hold two accumulator sets;
//   * every B fragment read from LDS feeds both row tiles (6 MFMAs per pair of ds_read_b128 in pass B),
//     halving the LDS read traffic of the 8-wave structure;
//   * the epilogue of unit u-1 (exp2 / sums / maxima / candidate test on the finished accumulator set)
//     is interleaved IN THE SAME WAVE between the MFMAs of unit u (other accumulator set): a few VALU
//     instructions per k-step ride in the shadow of that k-step's six 32-cycle MFMAs, instead of
//     relying on a second wave of the SIMD to fill the matrix pipe (measured: two co-resident waves
//     run their MFMA phases one after the other and the pipe idles ~50 % of the time).
#include "fm_internal.h"

namespace fm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef FM_PF_SUM
#define FM_PF_SUM 2
#endif
#ifndef FM_PF_MAX
#define FM_PF_MAX 4
#endif

struct CorrArgs {
  const _Float16* hi0; const _Float16* lo0; const _Float16* hi1; const _Float16* lo1;
  const float* nmr; const float* nmc;
  float* rowpart; float* colpart;
  int* cand_count; int* cand_j; float* cand_x; unsigned* flags;
  int L, S, Lp, Sp, panels, tiles, splits, tiles_per_split, slots;
  float k;    // log2(e) / (C*T): raw dot product -> log2-domain similarity
  float lt;   // log2(thr)
};

__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}

template <int C>
__device__ __forceinline__ int swz(int col) {
  constexpr int CHUNKS = C / 8;
  return CHUNKS >= 16 ? (col & 15) : ((col >> 1) & (CHUNKS - 1));
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int C, int MODE>
__global__ __launch_bounds__(256) void k_corr(CorrArgs a) {
  constexpr int KSTEPS = C / 16;
  constexpr int ROWB = C * 2;
  constexpr int CHUNKS = C / 8;
  constexpr int PLANES = MODE ? 2 : 1;
  constexpr int PLANE_BYTES = kTileCols * ROWB;
  constexpr int BUF_BYTES = PLANES * PLANE_BYTES;
  constexpr int INSTR_PER_WAVE = PLANE_BYTES / 1024 / 4;
  constexpr int EPK = 32 / KSTEPS;          // epilogue elements folded per k-step (32 per unit and lane)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  if (MODE == 2 && !(*a.flags & FM_INT_SCREEN_OVERFLOW)) return;   // uniform: fast screening sufficed
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;

  int kk = xcd_remap(blockIdx.x, gridDim.x);
  const int panel = kk % a.panels; kk /= a.panels;
  const int split = kk % a.splits;
  const int b = kk / a.splits;
  const int t0 = split * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.tiles);

  const _Float16* planes1[2] = {a.hi1 + (long)b * a.Sp * C, a.lo1 + (long)b * a.Sp * C};
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
#pragma unroll
      for (int n = 0; n < INSTR_PER_WAVE; ++n) {
        const int instr = wv * INSTR_PER_WAVE + n;
        const int byte = instr * 1024 + lane * 16;
        const int col = byte / ROWB;
        const int q = ((byte % ROWB) >> 4) ^ swz<C>(col);
        const _Float16* src = planes1[p] + (long)(t * kTileCols + col) * C + q * 8;
        glds16(src, smem + buf * BUF_BYTES + p * PLANE_BYTES + instr * 1024);
      }
    }
  };
  if (t0 < t1) stage(t0, 0);

  // ---- this wave's 64 rows as A fragments (fragment-major planes: 1 KiB per (32-row block, k-step)) ----
  const int wrow0 = panel * kPanelRows + wv * 64;
  half8 ahi[2][KSTEPS], alo[MODE ? 2 : 1][MODE ? KSTEPS : 1];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const long off = (((long)b * a.Lp + wrow0 + rt * 32) / 32 * KSTEPS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) ahi[rt][ks] = *reinterpret_cast<const half8*>(a.hi0 + off + ks * 512);
    if (MODE) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) alo[rt][ks] = *reinterpret_cast<const half8*>(a.lo0 + off + ks * 512);
    }
  }

  // row statistics / stabilisers, one per accumulator register: row = wrow0 + 32*rt + (g&3) + 8*(g>>2) + 4*h
  float rstat[2][16], nmr[2][16];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      rstat[rt][g] = MODE ? 0.f : -INFINITY;
      nmr[rt][g] = MODE ? a.nmr[(long)b * a.Lp + wrow0 + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h] : 0.f;
    }
  const bool row_edge = (wrow0 + 64 > a.L);     // wave-uniform: some of this wave's rows are padding

  const int lanebase0 = r * ROWB + (((h * (CHUNKS / 2)) ^ swz<C>(r)) << 4);              // columns 0..31 of a tile
  const int lanebase1 = (32 + r) * ROWB + (((h * (CHUNKS / 2)) ^ swz<C>(32 + r)) << 4);  // columns 32..63
  float* colout = a.colpart + (((long)b * a.panels + panel) * kColParts + wv) * a.Sp;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // ---- per-element epilogue: fold one accumulator value into the statistics ----
  struct Epi { float cst, best, nmc; };
  auto elem = [&](Epi& e, int rt, int g, float x) {      // x = -inf for padded rows / columns
    if (MODE == 0) {
      rstat[rt][g] = fmaxf(rstat[rt][g], x);
      e.cst = fmaxf(e.cst, x);
    } else {
      const float rr = __builtin_fmaf(x, a.k, nmr[rt][g]);
      const float cc = __builtin_fmaf(x, a.k, e.nmc);
      if (MODE == 1) {
        rstat[rt][g] += __builtin_amdgcn_exp2f(rr);
        e.cst += __builtin_amdgcn_exp2f(cc);
      }
      e.best = fmaxf(e.best, fminf(rr, cc));
    }
  };
  auto epi_begin = [&](Epi& e, int u) {
    e.cst = MODE ? 0.f : -INFINITY;
    e.best = -INFINITY;
    e.nmc = MODE ? a.nmc[(long)b * a.Sp + (u >> 1) * kTileCols + (u & 1) * 32 + r] : 0.f;
  };
  auto epi_end = [&](Epi& e, int u, const f32x16 (&acc)[2]) {
    const int col = (u >> 1) * kTileCols + (u & 1) * 32 + r;
    if (MODE != 2) {
      const float o = __shfl_xor(e.cst, 32);
      const float v = MODE ? e.cst + o : fmaxf(e.cst, o);
      if (h == 0) colout[col] = v;                       // this wave's 64 rows of column `col`
    }
    if (MODE && __any(e.best > a.lt)) {                  // rare: some lane holds a candidate in this unit
      int rbase = wrow0 + 4 * h;
      asm volatile("" : "+v"(rbase));                    // keep per-row addresses from being hoisted
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float x = acc[rt][g];
          const float rr = __builtin_fmaf(x, a.k, nmr[rt][g]);
          const float cc = __builtin_fmaf(x, a.k, e.nmc);
          const int row = rbase + rt * 32 + (g & 3) + 8 * (g >> 2);
          if (rr > a.lt && cc > a.lt && row < a.L && col < a.S) {
            const long grow = (long)b * a.Lp + row;
            const int pos = atomicAdd(&a.cand_count[grow], 1);
            if (pos < a.slots) { a.cand_j[grow * a.slots + pos] = col; a.cand_x[grow * a.slots + pos] = x; }
            else atomicOr(a.flags, MODE == 1 ? (unsigned)FM_INT_SCREEN_OVERFLOW : (unsigned)FM_DEV_CANDIDATES);
          }
        }
      }
    }
  };

  // ---- MFMAs of unit u into accN, with the epilogue of unit u-1 (held in accP) folded in between ----
  // B fragments are read PF k-steps ahead (inline asm + counted lgkmcnt, see coarse_corr.hip).
  auto run = [&](int u, f32x16 (&accN)[2], const f32x16 (&accP)[2], bool have_prev) {
    const unsigned base = lds0 + (((u >> 1) - t0) & 1) * BUF_BYTES + ((u & 1) ? lanebase1 : lanebase0);
    constexpr int PF = MODE ? FM_PF_SUM : FM_PF_MAX;
    constexpr int RING = PF + 1;
    half8 bh[RING], bl[MODE ? RING : 1];
    auto issue = [&](int ks) {
      const unsigned la = base ^ (unsigned)(ks << 4);
      asm volatile("ds_read_b128 %0, %1" : "=v"(bh[ks % RING]) : "v"(la));
      if (MODE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[ks % RING]) : "v"(la), "i"(PLANE_BYTES));
    };
    Epi e;
    if (have_prev) epi_begin(e, u - 1);
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
        else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bh[ks % RING]));
      }
      static_assert(PF * (MODE ? 2 : 1) <= 8 && (MODE || PF <= 4 || PF == 8), "lgkmcnt ladder");
      const half8 h8 = bh[ks % RING];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        if (ks == 0) {
          // accumulator starts at 0, or at -inf for padded rows / columns (no masks in the epilogue)
          f32x16 z;
#pragma unroll
          for (int g = 0; g < 16; ++g) z[g] = 0.f;
          const int ucol0 = (u >> 1) * kTileCols + (u & 1) * 32;
          if (row_edge || ucol0 + 32 > a.S) {
            const float cb = (ucol0 + r < a.S) ? 0.f : -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g)
              z[g] = (wrow0 + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h < a.L) ? cb : -INFINITY;
          }
          accN[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rt][ks], h8, z, 0, 0, 0);
        } else {
          accN[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rt][ks], h8, accN[rt], 0, 0, 0);
        }
        if (MODE) {
          accN[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[rt][ks], h8, accN[rt], 0, 0, 0);
          accN[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rt][ks], bl[ks % RING], accN[rt], 0, 0, 0);
        }
      }
      // a slice of the previous unit's epilogue rides in the shadow of this k-step's MFMAs
      if (have_prev) {
#pragma unroll
        for (int q = 0; q < EPK; ++q) {
          const int idx = ks * EPK + q;
          elem(e, idx >> 4, idx & 15, accP[idx >> 4][idx & 15]);
        }
      }
    }
    if (have_prev) epi_end(e, u - 1, accP);
  };

  f32x16 acc0[2], acc1[2];      // accumulator sets of the even / odd unit of a tile
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc0[rt][g] = 0.f; acc1[rt][g] = 0.f; }

  __syncthreads();              // first tile landed

  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) stage(t + 1, ((t - t0) & 1) ^ 1);
    run(2 * t, acc0, acc1, t > t0);        // overlaps the epilogue of unit 2t-1
    run(2 * t + 1, acc1, acc0, true);      // overlaps the epilogue of unit 2t
    __syncthreads();            // tile consumed by every wave; next tile landed (LDS-DMA drained)
  }
  if (t1 > t0) {                // epilogue of the last unit
    Epi e;
    epi_begin(e, 2 * t1 - 1);
#pragma unroll
    for (int idx = 0; idx < 32; ++idx) elem(e, idx >> 4, idx & 15, acc1[idx >> 4][idx & 15]);
    epi_end(e, 2 * t1 - 1, acc1);
  }

  if (MODE == 2) return;
  // ---- row statistics of this workgroup's column range: reduce over the 32 lanes of each half ----
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float v = rstat[rt][g];
#pragma unroll
      for (int m = 1; m <= 16; m <<= 1) {
        const float o = __shfl_xor(v, m);
        v = MODE ? v + o : fmaxf(v, o);
      }
      rstat[rt][g] = v;
    }
    if (r == 0) {
      float* out = a.rowpart + ((long)b * a.splits + split) * a.Lp + wrow0 + rt * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 16; ++g) out[(g & 3) + 8 * (g >> 2)] = rstat[rt][g];
    }
  }
}

template <int C, int MODE>
static hipError_t launch_corr_t(const CorrArgs& a, int blocks, hipStream_t st) {
  constexpr int BUF_BYTES = (MODE ? 2 : 1) * kTileCols * C * 2;
  constexpr int SMEM = 2 * BUF_BYTES;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_corr<C, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_corr<C, MODE>), dim3(blocks), dim3(256), SMEM, st, a);
  return hipGetLastError();
}

hipError_t launch_corr(int mode, const CoarseWs& w, char* base, float inv_ct, float thr, hipStream_t st, float* conf) {
  if (mode == 3 || conf) return hipErrorNotSupported;   // dense conf output only exists in the 8-wave sweep
  CorrArgs a;
  a.hi0 = (const _Float16*)(base + w.hi0); a.lo0 = (const _Float16*)(base + w.lo0);
  a.hi1 = (const _Float16*)(base + w.hi1); a.lo1 = (const _Float16*)(base + w.lo1);
  a.nmr = (const float*)(base + (mode == 2 ? w.nmr2 : w.nmr));
  a.nmc = (const float*)(base + (mode == 2 ? w.nmc2 : w.nmc));
  a.rowpart = (float*)(base + (mode ? w.rowB : w.rowA));
  a.colpart = (float*)(base + (mode ? w.colB : w.colA));
  a.cand_count = (int*)(base + w.cand_count); a.cand_j = (int*)(base + w.cand_j);
  a.cand_x = (float*)(base + w.cand_conf);   // raw dot product now, replaced by conf in k_cand_conf
  a.flags = (unsigned*)(base + w.scalars);
  a.L = w.L; a.S = w.S; a.Lp = w.Lp; a.Sp = w.Sp; a.panels = w.panels; a.tiles = w.tiles;
  a.splits = w.splits; a.tiles_per_split = (w.tiles + w.splits - 1) / w.splits; a.slots = w.slots;
  a.k = inv_ct * kLog2e;
  a.lt = log2f(thr) - (mode == 2 ? 2e-4f : 0.f);   // pass C compares rounded log-softmax values: small guard
  const int blocks = w.N * w.splits * w.panels;
#define FM_CORR_CASE(CC)                                                     \
  case CC: return mode == 2 ? launch_corr_t<CC, 2>(a, blocks, st)            \
                 : (mode ? launch_corr_t<CC, 1>(a, blocks, st) : launch_corr_t<CC, 0>(a, blocks, st));
  switch (w.C) {
    FM_CORR_CASE(64)
    FM_CORR_CASE(128)
    FM_CORR_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef FM_CORR_CASE
}

}  // namespace fm
