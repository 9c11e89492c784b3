// Fine stage: window crop (network/module/fine_preprocess.py:43-50) and the dual-direction
// local-window correlation + soft-argmax (network/utils/fine_matching_new.py:50-79).
#include "fm_device.h"
#include "fm_maps_device.h"

namespace fm {

// Workgroups are dealt round-robin over the 8 XCDs (private L2s).  Give each XCD a contiguous range
// of windows: matches are sorted by coarse cell, so neighbouring windows share cache lines of the
// fine map and one XCD then pulls only its part of the map over the fabric (speed only).
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}

// ----------------------------------------------------------------------------------------
// Window crop for NCHW maps with Cf = 64 (W compile-time): ONE WAVE PER WINDOW, four windows per
// workgroup, no workgroup barrier.
// The reference unfolds EVERY coarse cell's window (60 MB per image at 640x480) and then selects M
// of them; here only the M selected windows are read, straight from the NCHW map.  Lanes run along
// (channel, window row, x) with x fastest, so one wave-load touches ~64/W short runs in 2-3 channel
// planes.  (Lane = channel would put the 64 lanes of a load one plane stride apart - 307200 B at
// 240x320 - which lands on a handful of L2/HBM channels; measured 2x slower.)  All loads of the window
// are issued before the first use (25 / 49 in flight per lane).  The [c][r] -> [r][c] transpose goes
// through a wave-private LDS tile (pitch 68 floats: 16-byte aligned rows) and the window leaves as one
// contiguous record in 16-byte stores.  Window origin = stride*cell - pad with the reference's literal
// pad = 2, zero outside the map.
// (One workgroup per window, 3776 four-wave workgroups at 640x480, was bound by workgroup launch
// rate: its time followed the number of workgroups, not the bytes.)
// ----------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// In-place context merge of one window tile in LDS (fine_preprocess.py:56-59, the window half of
// merge_feat): tile[r][n] <- sum_k W_w[n][k] * tile[r][k] + ctx[n], r = window position, k/n = channels.
// (The context half W_c . down_proj(feat_c) + bias does not depend on the position: it arrives as ctx, one
// row of a per-cell table the caller computes with a plain library GEMM.)  f32-equivalent product on the f16
// matrix cores, as in the coarse stage: window values and weights are split x = hi + lo (f16 each) and
// hi*hi + lo*hi + hi*lo is accumulated in f32 (~2^-22 relative).  wpack = the weights pre-split into MFMA B
// fragments by fm_merge_pack_weights: [n-tile 2][k-step 4][hi|lo][lane 64] x 8 halfs.
// Both operands carry exact power-of-two scales, undone on the accumulator: the matrix cores flush float16 SUBNORMAL
// inputs, and the lo half of anything below 2^-3 - every weight of a 128-wide Linear layer - would be one (see
// fine_tf.hip).  Weights x kMergeWgtScale (fixed: |w| < 16, merge_feat is a 128-wide Linear layer).  Window values x a
// scale that FOLLOWS THE WINDOW (round 4): the power of two that brings the window's largest magnitude into
// [2^13, 2^14) - a fixed 2^8 turned every window value beyond 255.9 into inf without a word (un-normalised backbone
// features; found by a matcher test on maps of magnitude 3e5).
constexpr float kMergeWgtScale = 4096.f;
// exact power of two s with amax * s in [2^13, 2^14), clamped to [2^-100, 2^40] (a window of zeros / denormals: the
// context term alone decides, and c * s * 2^12 must stay inside float32)
__device__ __forceinline__ float merge_act_scale(float amax) {
  const unsigned bits = __float_as_uint(amax);
  const int e = (int)((bits >> 23) & 0xffu);           // amax in [2^(e-127), 2^(e-126))
  int k = 140 - e;                                     // 13 - (e - 127)
  if (e == 255) k = 0;                                 // (Inf / NaN window: the result is NaN either way)
  k = k > 40 ? 40 : (k < -100 ? -100 : k);
  return __uint_as_float((unsigned)(127 + k) << 23);
}
template <int W>
__device__ __forceinline__ void wave_merge_tile(float* tile, int lane, const half8* __restrict__ wpack,
                                                const float* __restrict__ ctx_row) {
  constexpr int WW = W * W, PITCH = 68, MT = (WW + 31) / 32;
  const int r = lane & 31, h = lane >> 5;
  const float c0 = ctx_row[r], c1 = ctx_row[32 + r];
  half8 bhi[2][4], blo[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bhi[nt][ks] = wpack[((nt * 4 + ks) * 2 + 0) * 64 + lane];
      blo[nt][ks] = wpack[((nt * 4 + ks) * 2 + 1) * 64 + lane];
    }
  // the window's largest magnitude (every lane reads the values it will split below: the 64 lanes cover the tile)
  float amax = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(32 * mt + r, WW - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 p = *reinterpret_cast<const float4*>(tile + row * PITCH + ks * 16 + 8 * h);
      const float4 q = *reinterpret_cast<const float4*>(tile + row * PITCH + ks * 16 + 8 * h + 4);
      amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), fabsf(p.w))),
                               fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)))));
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
  const float act_scale = merge_act_scale(amax);
  // A fragments of all window rows first (the tile is overwritten below): lane (r, h) holds row r,
  // channels 16*ks + 8*h .. +7; rows beyond the window re-read its last row (their outputs are dropped)
  half8 ahi[MT][4], alo[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(32 * mt + r, WW - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 p = *reinterpret_cast<const float4*>(tile + row * PITCH + ks * 16 + 8 * h);
      const float4 q = *reinterpret_cast<const float4*>(tile + row * PITCH + ks * 16 + 8 * h + 4);
      const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xs = x[e] * act_scale;
        const _Float16 hh = (_Float16)xs;
        ahi[mt][ks][e] = hh;
        alo[mt][ks][e] = (_Float16)(xs - (float)hh);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const float inv_scale = (1.0f / act_scale) * (1.0f / kMergeWgtScale);      // (both exact powers of two)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = (nt ? c1 : c0) * (act_scale * kMergeWgtScale);   // column n = 32*nt + r
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mt][ks], bhi[nt][ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mt][ks], bhi[nt][ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mt][ks], blo[nt][ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 32 * mt + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (row < WW) tile[row * PITCH + 32 * nt + r] = acc[g] * inv_scale;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// One W x W window of a 64-channel NCHW map into a wave-private LDS tile [position][channel] (pitch 68 floats).
// lane -> (channel within the instruction, window position) is the same for every load: W = 5 puts two channels x 32
// position slots (25 used) in one wave-load, W = 7 one channel x 64 slots (49 used).  Only the channel advances from
// load to load, by a SCALAR offset, so the address math is done once.  Buffer loads through a descriptor of this
// sample's map (src is wave-uniform): positions outside the map (the zero padding of the unfold) get an out-of-range
// offset, which the hardware range check answers with 0 - no branch, no 64-bit address per load.
template <int W>
__device__ __forceinline__ void wave_load_window64(const float* src, int Hf, int Wf, int oy, int ox, float* tile, int lane) {
  constexpr int CF = 64, WW = W * W, PITCH = CF + 4;
  constexpr int SLOTS = W == 5 ? 32 : 64, CPI = 64 / SLOTS, NLOAD = CF / CPI;
  static_assert(W == 5 || W == 7, "position decode below is for W in {5,7}");
  const int rem = lane & (SLOTS - 1), hi = lane / SLOTS;
  const int wy = W == 5 ? (rem * 13) >> 6 : (rem * 37) >> 8;       // rem / W for rem < W*W
  const int wx = rem - wy * W;
  const int y = oy + wy, x = ox + wx;
  const bool ok = rem < WW && y >= 0 && y < Hf && x >= 0 && x < Wf;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, CF * Hf * Wf * 4, 0x00020000);
  const unsigned voff = ok ? (unsigned)((hi * Hf + y) * Wf + x) * 4u : 0x80000000u;
  const int step = Hf * Wf * 4 * CPI;                               // bytes from one load to the next
  float v[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
#ifndef FM_ABL_G_NOLOAD      // timing-only ablations (wrong results): never defined in the shipped build
    v[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, it * step, 0));
#else
    v[it] = 0.f;
#endif
  }
  float* slot = tile + rem * PITCH + hi;
#pragma unroll
  for (int it = 0; it < NLOAD; ++it)
    if (rem < WW) slot[it * CPI] = v[it];
  __builtin_amdgcn_wave_barrier();       // same-wave LDS accesses are processed in order: no s_barrier needed
}

template <int W, bool MERGE = false>
__device__ __forceinline__ void wave_copy_window64(const float* src, int Hf, int Wf, int oy, int ox,
                                                   float* __restrict__ dst, float* tile, int lane,
                                                   const half8* __restrict__ wpack = nullptr,
                                                   const float* __restrict__ ctx_row = nullptr) {
  constexpr int CF = 64, WW = W * W, TOTAL = CF * WW, PITCH = CF + 4;
  wave_load_window64<W>(src, Hf, Wf, oy, ox, tile, lane);
  if (MERGE) wave_merge_tile<W>(tile, lane, wpack, ctx_row);
  float4* dst4 = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int it = 0; it < (TOTAL / 4 + 63) / 64; ++it) {
    const int idx = it * 64 + lane;          // float4 index: position idx / 16, channels 4*(idx % 16) ..
    if (idx < TOTAL / 4) {
      const float4 q = *reinterpret_cast<const float4*>(tile + (idx >> 4) * PITCH + (idx & 15) * 4);
#ifdef FM_ABL_G_NOSTORE
      if (q.x == 1.2345e-30f) dst4[idx] = q;
#else
      dst4[idx] = q;
#endif
    }
  }
  __builtin_amdgcn_wave_barrier();       // the tile may be refilled by this wave right away
}

constexpr int kGatherTileFloats(int W) { return W * W * 68; }

// list order: window m of the match list (any ids; the generic entry point)
template <int W, bool MERGE>
__global__ __launch_bounds__(256) void k_gather_nchw64(const float* __restrict__ feat, int Hf, int Wf, int stride,
                                                       int pad, int w_c, const int64_t* __restrict__ b_ids,
                                                       const int64_t* __restrict__ ids,
                                                       const int32_t* __restrict__ d_count, int m_max,
                                                       float* __restrict__ out, const half8* __restrict__ wpack,
                                                       const float* __restrict__ ctx, int ctx_cells) {
  constexpr int CF = 64, TOTAL = CF * W * W;
  __shared__ __attribute__((aligned(16))) float tile[4 * kGatherTileFloats(W)];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  // spread the M live windows (not the m_max launched ones) contiguously over the XCDs
  const int per = (M + 7) >> 3;
  const int slot = (int)(blockIdx.x >> 3) * 4 + wv;
  const int m = (blockIdx.x & 7) * per + slot;
  if (slot >= per || m >= M) return;
  const int b = __builtin_amdgcn_readfirstlane((int)b_ids[m]);       // wave-uniform values into scalars
  const int id = __builtin_amdgcn_readfirstlane((int)ids[m]);
  const int cy = id / w_c;
  wave_copy_window64<W, MERGE>(feat + (long)b * CF * Hf * Wf, Hf, Wf, cy * stride - pad, (id - cy * w_c) * stride - pad,
                               out + (long)m * TOTAL, tile + wv * kGatherTileFloats(W), lane, wpack,
                               MERGE ? ctx + ((long)b * ctx_cells + id) * CF : nullptr);
}

// ----------------------------------------------------------------------------------------
// k_gather_cellorder64<W>: the same per-window copy, but the waves run over the CELLS of this image
// in raster order (cell -> match map from the coarse stage, 0 = unmatched) and each XCD takes a
// contiguous band of cells.  For image 1 the match list is sorted by the image-0 cell, so in list
// order the windows of one XCD are scattered over the whole map (20 MB at 640x480 against a 4 MB L2:
// every 20-byte run misses to the fabric); in cell order one XCD touches only its band (2.5 MB).
// Matches that lost their cell to an exactly tied match (listed by k_emit) are copied by the first
// waves of the grid, one each; if that list overflowed (> kTieCap) every wave scans its slice of the
// match list instead.
// ----------------------------------------------------------------------------------------
// One launch can serve BOTH images of the pairs (blockIdx.y = image): the two crops are independent and each
// is bound by one round of memory latency at 640x480, so a second launch only adds its ramp and tail.
struct CellImage {
  const float* feat; int Hf, Wf, w_c, cells, total_cells;
  const int32_t* cell_to_match; int cell_pitch; const int32_t* ties;
  const int64_t* ids; float* out; const float* ctx;
};
struct CellArgs {
  CellImage im[2];
  int stride, pad, m_max;
  const int64_t* b_ids; const int32_t* d_count; const half8* wpack;
};

template <int W, bool MERGE>
__global__ __launch_bounds__(256) void k_gather_cellorder64(CellArgs a) {
  constexpr int CF = 64, TOTAL = CF * W * W;
  __shared__ __attribute__((aligned(16))) float tile_all[4 * kGatherTileFloats(W)];
  const CellImage& I = a.im[blockIdx.y];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = tile_all + wv * kGatherTileFloats(W);
  const int M = a.d_count ? min(a.d_count[0], a.m_max) : a.m_max;
  const int nties = I.ties[0];
  const int lid = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wv;      // grid.x is a multiple of 8
  // job = (match, sample, cell) this wave copies next; first its own cell, then tie losers (rare)
  int jm = -1, jb = 0, jcell = 0;
  if (lid < I.total_cells) {
    const int b = lid / I.cells, cell = lid - b * I.cells;
    const int m = I.cell_to_match[(long)b * I.cell_pitch + cell] - 1;
    if (m >= 0 && m < M) { jm = m; jb = b; jcell = cell; }
  }
  const int gwave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
  const int mode = nties == 0 ? 0 : (nties <= kTieCap ? 1 : 2);          // 2: list overflowed, scan the matches
  int q = gwave;
  const int chunk = (M + nwaves - 1) / nwaves;
  int scan = gwave * chunk;
  const int scan_end = min(M, scan + chunk);
  for (;;) {
    if (jm >= 0) {
      jm = __builtin_amdgcn_readfirstlane(jm); jb = __builtin_amdgcn_readfirstlane(jb);
      jcell = __builtin_amdgcn_readfirstlane(jcell);
      const int cy = jcell / I.w_c;
      wave_copy_window64<W, MERGE>(I.feat + (long)jb * CF * I.Hf * I.Wf, I.Hf, I.Wf, cy * a.stride - a.pad,
                                   (jcell - cy * I.w_c) * a.stride - a.pad, I.out + (long)jm * TOTAL, tile, lane,
                                   a.wpack, MERGE ? I.ctx + ((long)jb * I.cells + jcell) * CF : nullptr);
    }
    jm = -1;
    if (mode == 0) break;                                                // the common case: no exact ties
    if (mode == 1) {
      if (q >= nties) break;
      const int m = I.ties[1 + q];
      q += nwaves;
      if (m >= 0 && m < M) { jm = m; jb = (int)a.b_ids[m]; jcell = (int)I.ids[m]; }
    } else {
      if (scan >= scan_end) break;
      const int m = scan++;
      const int mb = (int)a.b_ids[m], id = (int)I.ids[m];
      if (I.cell_to_match[(long)mb * I.cell_pitch + id] != m + 1) { jm = m; jb = mb; jcell = id; }
    }
  }
}

// element of a float32 / float16 / bfloat16 map as float32 (exact)
template <int DT> struct MapElem { using type = float; };
template <> struct MapElem<FM_F16> { using type = unsigned short; };
template <> struct MapElem<FM_BF16> { using type = unsigned short; };
template <int DT>
__device__ __forceinline__ float map_value(const typename MapElem<DT>::type* p, long i) {
  if constexpr (DT == FM_F32) return p[i];
  else return half_bits_to_float(p[i], DT);
}

// generic fallback (any Cf / W, any element type): one workgroup per window, transposed through LDS
template <int DT>
__global__ __launch_bounds__(256) void k_gather_nchw(const typename MapElem<DT>::type* __restrict__ feat, int Cf, int Hf, int Wf, int W,
                                                     int stride, int pad, int w_c, const int64_t* __restrict__ b_ids,
                                                     const int64_t* __restrict__ ids, const int32_t* __restrict__ d_count,
                                                     int m_max, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [WW][Cf+1]
  const int m = blockIdx.x;
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if (m >= M) return;
  const int WW = W * W;
  const int b = (int)b_ids[m];
  const int id = (int)ids[m];
  const int oy = (id / w_c) * stride - pad;
  const int ox = (id % w_c) * stride - pad;
  const typename MapElem<DT>::type* src = feat + (long)b * Cf * Hf * Wf;
  const int total = Cf * WW;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int c = idx / WW;
    const int rem = idx - c * WW;
    const int wy = rem / W, wx = rem - wy * W;
    const int y = oy + wy, x = ox + wx;
    float v = 0.f;
    if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = map_value<DT>(src, ((long)c * Hf + y) * Wf + x);
    tile[rem * (Cf + 1) + c] = v;
  }
  __syncthreads();
  float* dst = out + (long)m * total;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int rpos = idx / Cf;
    const int c = idx - rpos * Cf;
    dst[idx] = tile[rpos * (Cf + 1) + c];
  }
}

// NHWC storage: every window row is W*Cf contiguous elements - a straight copy, four channels per thread and step.
template <int DT>
__global__ __launch_bounds__(256) void k_gather_nhwc(const typename MapElem<DT>::type* __restrict__ feat, int Cf, int Hf, int Wf, int W,
                                                     int stride, int pad, int w_c, const int64_t* __restrict__ b_ids,
                                                     const int64_t* __restrict__ ids, const int32_t* __restrict__ d_count,
                                                     int m_max, float* __restrict__ out) {
  const int m = blockIdx.x;
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if (m >= M) return;
  const int WW = W * W;
  const int b = (int)b_ids[m];
  const int id = (int)ids[m];
  const int oy = (id / w_c) * stride - pad;
  const int ox = (id % w_c) * stride - pad;
  const int c4 = Cf / 4;
  const typename MapElem<DT>::type* src = feat + (long)b * Hf * Wf * Cf;
  float4* dst = reinterpret_cast<float4*>(out + (long)m * WW * Cf);
  for (int idx = threadIdx.x; idx < WW * c4; idx += 256) {
    const int rpos = idx / c4;
    const int c = idx - rpos * c4;
    const int wy = rpos / W, wx = rpos - wy * W;
    const int y = oy + wy, x = ox + wx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < Hf && x >= 0 && x < Wf) {
      const long e = ((long)y * Wf + x) * c4 + c;          // in units of four elements
      if constexpr (DT == FM_F32) v = reinterpret_cast<const float4*>(src)[e];
      else v = half4_to_float4(reinterpret_cast<const uint2*>(src)[e], DT);
    }
    dst[idx] = v;
  }
}

// ----------------------------------------------------------------------------------------
// k_fine: one wave per match, lane = channel (Cf = 64).
//   q0[c]   = b0 + sum_r w0[r] F0[r][c]            (Linear over the POSITION axis, :50,53)
//   sim0[r] = sum_c q0[c] F1[r][c]                 (:56)   -> refines keypoint 0
//   q1 / sim1 symmetrically (:51,54,57)            -> refines keypoint 1
//   heat = softmax(sim / sqrt(C)); coords = E[grid]; std = sum sqrt(clamp(var, 1e-10))  (:58-73)
//   out  = mkpts_c + (coords * (W//2) * scale + W//2), std                              (:75-79)
// The WW cross-lane sums of sim are done with one butterfly "transpose-reduce" built from DPP and
// permlane swaps (no LDS): after the exchange steps every lane holds one sim[pos], so the softmax
// runs with lane = window position.
// ----------------------------------------------------------------------------------------
// Cross-lane exchange without LDS: v[lane ^ MASK] via DPP (1, 2, 4, 8) or the gfx950 permlane swaps
// (16, 32), folded straight into the reduction operator.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xf, BANK, false));
}
struct OpAdd { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };

// op(v[lane], v[lane ^ MASK]) in every lane
template <int MASK, class Op>
__device__ __forceinline__ float pair_op(float v) {
  if constexpr (MASK == 1) return Op::f(v, dpp_mov<0xB1, 0xf>(v, v));           // quad_perm [1,0,3,2]
  else if constexpr (MASK == 2) return Op::f(v, dpp_mov<0x4E, 0xf>(v, v));      // quad_perm [2,3,0,1]
  else if constexpr (MASK == 4) {
    float t = dpp_mov<0x104, 0x5>(v, v);      // row_shl:4 into banks 0,2 (lanes with bit 2 clear read lane+4)
    t = dpp_mov<0x114, 0xA>(t, v);            // row_shr:4 into banks 1,3 (lanes with bit 2 set read lane-4)
    return Op::f(v, t);
  } else if constexpr (MASK == 8) return Op::f(v, dpp_mov<0x128, 0xf>(v, v));   // row_ror:8
  else if constexpr (MASK == 16) {
    // v_permlane16_swap a, b: odd rows of a <-> even rows of b.  With a = b = v: a = {r0,r0,r2,r2},
    // b = {r1,r1,r3,r3}, so op(a, b) is the pair result in every lane.  Inline asm because hipcc
    // (ROCm 7.2) returns the first result twice from the builtin when both operands are one value;
    // s_nop 1 = the two wait states between a VALU write of an operand and the swap.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return Op::f(a, b);
  } else {
    float a = v, b = v;   // lanes 32-63 of a <-> lanes 0-31 of b: a = {lo,lo}, b = {hi,hi}
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return Op::f(a, b);
  }
}

template <class Op>
__device__ __forceinline__ float wave_all(float v) {
  v = pair_op<1, Op>(v); v = pair_op<2, Op>(v); v = pair_op<4, Op>(v);
  v = pair_op<8, Op>(v); v = pair_op<16, Op>(v); v = pair_op<32, Op>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_all<OpAdd>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_all<OpMax>(v); }

// Transpose-reduce: p[k] = this lane's term of sum k (k < NP = 32 or 64) -> every sum ends in ONE lane (NP = 64) or in
// a pair of neighbouring lanes (NP = 32), tr_index(lane) tells which.  Each level halves the registers: a lane keeps one
// half and hands the other to its partner.  Written so that the levels with many pairs cost TWO instructions per pair
// and no select (round 2's butterfly spent five: two DPP moves, two adds, one select):
//   lane ^ 32 : v_permlane32_swap X, Y leaves {X.lo, Y.lo} / {X.hi, Y.hi}; X + Y = X's sum in lanes 0-31, Y's in 32-63
//   lane ^ 16 : v_permlane16_swap likewise for the rows of 16 lanes
//   15 - i    : row_mirror DPP add with bank-masked writes: banks 0-1 keep X, banks 2-3 receive Y's sum (into X)
//   7 - i     : row_half_mirror, banks 0 / 2 keep X, banks 1 / 3 receive Y's
//   3 - i     : inside the quad by select + quad_perm (one or two pairs are left by then)
//   i ^ 1     : NP = 64: one more transposing level; NP = 32: a plain sum (both lanes of a pair hold it)
// (s_nop 1: the two wait states a DPP / permlane operand needs after the VALU write of its register.)
template <int NP>
__device__ __forceinline__ float transpose_reduce(float (&p)[NP], int lane) {
  constexpr int H1 = NP / 2, H2 = NP / 4, H3 = NP / 8, H4 = NP / 16, H5 = NP / 32;
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < H1; ++k) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p[k]), "+v"(p[k + H1]));
#pragma unroll
  for (int k = 0; k < H1; ++k) p[k] += p[k + H1];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < H2; ++k) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p[k]), "+v"(p[k + H2]));
#pragma unroll
  for (int k = 0; k < H2; ++k) p[k] += p[k + H2];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < H3; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(p[k]));
#pragma unroll
  for (int k = 0; k < H3; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(p[k]) : "v"(p[k + H3]));
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < H4; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(p[k]));
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < H4; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(p[k]) : "v"(p[k + H4]));
  const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
#pragma unroll
  for (int k = 0; k < H5; ++k) {                 // lane 3 - i of the quad: quad_perm [3,2,1,0]
    const float own = b1 ? p[k + H5] : p[k], other = b1 ? p[k] : p[k + H5];
    p[k] = own + dpp_mov<0x1B, 0xf>(other, other);
  }
  if constexpr (NP == 64) {                      // lane i ^ 1 takes the second of the last two sums
    const float own = b0 ? p[1] : p[0], other = b0 ? p[0] : p[1];
    return own + dpp_mov<0xB1, 0xf>(other, other);
  } else {
    return p[0] + dpp_mov<0xB1, 0xf>(p[0], p[0]);
  }
}
// lane bits b5..b0 -> the sum it holds
template <int NP>
__device__ __forceinline__ int tr_index(int lane) {
  const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1;
  if constexpr (NP == 64) return 32 * b5 + 16 * b4 + 8 * b3 + 4 * b2 + 2 * b1 + (lane & 1);
  else return 16 * b5 + 8 * b4 + 4 * b3 + 2 * b2 + b1;
}

// Soft-argmax of BOTH directions at once.  pos = window position this lane holds (sim0 / sim1 are that position's
// similarities), on = lane takes part.  Per direction: heat = softmax(sim / sqrt(C)), expectation and variance of the
// normalised grid coordinates under it (kornia spatial_expectation2d on create_meshgrid) -> five sums each
// (sum e, sum gx e, sum gy e, sum gx^2 e, sum gy^2 e); the ten sums go through ONE partial transpose-reduce (16 values,
// four two-instruction levels, then two plain levels) instead of ten 6-level butterflies.
template <int W>
__device__ __forceinline__ void soft_argmax2(float sim0, float sim1, int pos, bool on, int lane, float inv_sqrt_c,
                                             float scale_f, float k0x, float k0y, float k1x, float k1y, float* out0,
                                             float* out1) {
  const float x0 = on ? sim0 * inv_sqrt_c : -INFINITY, x1 = on ? sim1 * inv_sqrt_c : -INFINITY;
  const float m0 = wave_max(x0), m1 = wave_max(x1);
  const float e0 = on ? __expf(x0 - m0) : 0.f, e1 = on ? __expf(x1 - m1) : 0.f;
  const int wy = pos / W, wx = pos - wy * W;
  const float gx = ((float)wx / (float)(W - 1) - 0.5f) * 2.f;     // kornia create_meshgrid, normalised
  const float gy = ((float)wy / (float)(W - 1) - 0.5f) * 2.f;
  float q[16] = {e0, gx * e0, gy * e0, gx * gx * e0, gy * gy * e0, e1, gx * e1, gy * e1, gx * gx * e1, gy * gy * e1,
                 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(q[k]), "+v"(q[k + 8]));
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] += q[k + 8];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(q[k]), "+v"(q[k + 4]));
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] += q[k + 4];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 2; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(q[k]));
#pragma unroll
  for (int k = 0; k < 2; ++k)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(q[k]) : "v"(q[k + 2]));
  asm volatile("s_nop 1" ::: "memory");
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(q[0]));
  asm volatile("s_nop 1" ::: "memory");
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(q[0]) : "v"(q[1]));
  float t = q[0];                                    // lane bits b5 b4 b3 b2 -> sum 8 b5 + 4 b4 + 2 b3 + b2; fold b1, b0
  // (hipcc's hazard recogniser does not see the asm above as the vector write it is: the wait states the DPP read
  // below needs are given by hand)
  asm volatile("s_nop 1" : "+v"(t));
  t += dpp_mov<0x4E, 0xf>(t, t);                     // lane ^ 2
  t += dpp_mov<0xB1, 0xf>(t, t);                     // lane ^ 1
  // (the sums are read from OTHER lanes below: without this hipcc sinks the last add into the lane-0 branch, where only
  // lane 0 executes it)
  asm volatile("" : "+v"(t));
  auto sum_k = [&](int k) {                          // (wave-uniform: a scalar)
    const int src = 32 * ((k >> 3) & 1) + 16 * ((k >> 2) & 1) + 8 * ((k >> 1) & 1) + 4 * (k & 1);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), src));
  };
  if (lane == 0) {
    float* const outs[2] = {out0, out1};
    const float kx[2] = {k0x, k1x}, ky[2] = {k0y, k1y};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float inv = 1.0f / sum_k(5 * d);
      const float cx = sum_k(5 * d + 1) * inv, cy = sum_k(5 * d + 2) * inv;
      const float vx = sum_k(5 * d + 3) * inv - cx * cx, vy = sum_k(5 * d + 4) * inv - cy * cy;
      outs[d][0] = kx[d] + (cx * (float)(W / 2) * scale_f + (float)(W / 2));
      outs[d][1] = ky[d] + (cy * (float)(W / 2) * scale_f + (float)(W / 2));
      outs[d][2] = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
    }
  }
}

// the arithmetic of one match once both windows sit in registers (lane = channel, f[r] = window position r)
template <int W>
__device__ __forceinline__ void fine_core(const float (&f0)[W * W], const float (&f1)[W * W], int lane,
                                          const float* __restrict__ mix0, const float* __restrict__ mix1, float k0x,
                                          float k0y, float k1x, float k1y, float scale_f, float* __restrict__ out0,
                                          float* __restrict__ out1) {
  constexpr int WW = W * W;
  constexpr int CF = 64;
  float q0 = mix0[WW], q1 = mix1[WW];
#pragma unroll
  for (int r = 0; r < WW; ++r) { q0 = __builtin_fmaf(mix0[r], f0[r], q0); q1 = __builtin_fmaf(mix1[r], f1[r], q1); }

#ifdef FM_ABL_F_NOCOMPUTE     // timing-only: loads and one store, no correlation / soft-argmax
  if (q0 + q1 == 1.2345e-30f) out0[0] = q0;
  return;
#endif
  const float inv_sqrt_c = 1.0f / sqrtf((float)CF);
  constexpr int NP = WW > 32 ? 64 : 32;      // butterfly width
  float p[NP];
#pragma unroll
  for (int r = 0; r < NP; ++r) p[r] = r < WW ? q0 * f1[r] : 0.f;
  const float sim0 = transpose_reduce<NP>(p, lane);
#pragma unroll
  for (int r = 0; r < NP; ++r) p[r] = r < WW ? q1 * f0[r] : 0.f;
  const float sim1 = transpose_reduce<NP>(p, lane);

  const int pos = tr_index<NP>(lane);
  const bool on = pos < WW && (NP == 64 || !(lane & 1));      // NP = 32: the even lane of the pair that holds a sum
  soft_argmax2<W>(sim0, sim1, pos, on, lane, inv_sqrt_c, scale_f, k0x, k0y, k1x, k1y, out0, out1);
}

template <int W>
__global__ __launch_bounds__(256) void k_fine(const float* __restrict__ win0, const float* __restrict__ win1, int m_max,
                                              const int32_t* __restrict__ d_count, const float* __restrict__ mix0,
                                              const float* __restrict__ mix1, const float* __restrict__ kc0,
                                              const float* __restrict__ kc1, float scale_f, float* __restrict__ out0,
                                              float* __restrict__ out1) {
  constexpr int WW = W * W;
  constexpr int CF = 64;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  if (m >= M) return;
  const float* p0 = win0 + (long)m * WW * CF + lane;
  const float* p1 = win1 + (long)m * WW * CF + lane;
  float f0[WW], f1[WW];
  // read-once data: non-temporal loads keep the windows from displacing the fine maps in L2 (-2 us per pair)
#pragma unroll
  for (int r = 0; r < WW; ++r) { f0[r] = __builtin_nontemporal_load(p0 + r * CF); f1[r] = __builtin_nontemporal_load(p1 + r * CF); }
  fine_core<W>(f0, f1, lane, mix0, mix1, kc0[m * 2], kc0[m * 2 + 1], kc1[m * 2], kc1[m * 2 + 1], scale_f,
               out0 + (long)m * 3, out1 + (long)m * 3);
}

// ----------------------------------------------------------------------------------------
// Channels-last maps: storage [N, Hf, Wf, 64] of the logical [N, 64, Hf, Wf] tensor (layout 1).  A window row is
// W x 256 contiguous bytes there - ten cache lines at W = 5 instead of the 64 x 20-byte runs of the NCHW layout - so
//   * the crop is a plain copy in 16-byte chunks (the output window [WW][64] is the W row segments one after the
//     other), one wave per window, every load of the window in flight before the first store;
//   * the fine stage can take its windows straight from the maps: k_fine_maps reads position r of both windows as one
//     coalesced 256-byte wave load each (exactly the loads k_fine issues on a window tensor) and the window tensors
//     (48 MB written and read back per 640x480 pair) never exist.  Matches are visited in list order, an XCD a
//     contiguous range: image-0 windows then walk the map in raster order, image-1 windows land wherever their partner
//     is - at ten full lines per window that costs nothing extra.
//   * NCHW maps get there through k_nchw_to_nhwc64 (a tiled transpose through LDS, both sides coalesced) into a
//     caller-provided scratch buffer.
// ----------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_gather_nhwc64(const float* __restrict__ feat, int Hf, int Wf, int stride, int pad,
                                                       int w_c, const int64_t* __restrict__ b_ids,
                                                       const int64_t* __restrict__ ids, const int32_t* __restrict__ d_count,
                                                       int m_max, float* __restrict__ out) {
  constexpr int ROW16 = W * 16;              // 16-byte chunks per window row (64 channels x 4 bytes = 16 chunks per pixel)
  constexpr int TOTAL16 = W * ROW16;         // ... per window: 400 / 784
  constexpr int NIT = (TOTAL16 + 63) / 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  const int per = (M + 7) >> 3;
  const int slot = (int)(blockIdx.x >> 3) * 4 + wv;
  const int m = (blockIdx.x & 7) * per + slot;
  if (slot >= per || m >= M) return;
  const int b = __builtin_amdgcn_readfirstlane((int)b_ids[m]);
  const int id = __builtin_amdgcn_readfirstlane((int)ids[m]);
  const int cy = id / w_c;
  const int oy = cy * stride - pad, ox = (id - cy * w_c) * stride - pad;
  const float4* src = reinterpret_cast<const float4*>(feat + (long)b * Hf * Wf * 64);
  float4* dst = reinterpret_cast<float4*>(out + (long)m * W * W * 64);
  float4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * 64 + lane;
    const int wy = k / ROW16, within = k - wy * ROW16;
    const int y = oy + wy, x = ox + (within >> 4);
    const bool ok = k < TOTAL16 && y >= 0 && y < Hf && x >= 0 && x < Wf;
    // (clamped address + select: no load behind a branch)
    const float4 t = src[ok ? ((long)y * Wf + x) * 16 + (within & 15) : 0];
    v[it] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * 64 + lane;
    if (k < TOTAL16) dst[k] = v[it];
  }
}

// NCHW0: image 0 is still in the reference's NCHW layout - its windows are visited in raster order here (the match list
// is sorted by the image-0 cell), which is the access pattern the NCHW loader of the cell-ordered crop copes with, so only
// image 1 (whose windows land wherever the partners are) needs the channels-last copy.
// DT = element type of the maps (FM_F32; FM_F16 / FM_BF16: channels-last maps of 2-byte elements, every value exact in
// float32, so the arithmetic - and the result - equals the float32 call on the up-cast maps).
template <int W, bool NCHW0, int DT = FM_F32>
__global__ __launch_bounds__(256) void k_fine_maps(const float* __restrict__ map0, const float* __restrict__ map1, int Hf0,
                                                   int Wf0, int Hf1, int Wf1, int stride, int pad, int w0c, int w1c,
                                                   const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                                   const int64_t* __restrict__ j_ids, const int32_t* __restrict__ d_count,
                                                   int m_max, const float* __restrict__ mix0, const float* __restrict__ mix1,
                                                   const float* __restrict__ kc0, const float* __restrict__ kc1, float scale_f,
                                                   float* __restrict__ out0, float* __restrict__ out1) {
  constexpr int WW = W * W;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  const int per = (M + 7) >> 3;
  __shared__ __attribute__((aligned(16))) float tile_all[NCHW0 ? 4 * kGatherTileFloats(W) : 4];
  // (a grid smaller than the match list walks it in strides: fm_fine_match_maps may cap the grid so that the kernel's
  // workgroups do not fill every wave slot of the chip while another pair's coarse kernels wait for theirs)
  for (int slot = (int)(blockIdx.x >> 3) * 4 + wv; slot < per; slot += (int)(gridDim.x >> 3) * 4) {
  const int m = (blockIdx.x & 7) * per + slot;
  if (m >= M) break;
  const int b = __builtin_amdgcn_readfirstlane((int)b_ids[m]);
  const int i = __builtin_amdgcn_readfirstlane((int)i_ids[m]);
  const int j = __builtin_amdgcn_readfirstlane((int)j_ids[m]);
  const int cy0 = i / w0c, cy1 = j / w1c;
  const int oy0 = cy0 * stride - pad, ox0 = (i - cy0 * w0c) * stride - pad;
  const int oy1 = cy1 * stride - pad, ox1 = (j - cy1 * w1c) * stride - pad;
  // One buffer descriptor per window ROW (base = that row of the map, size = one map row, or 0 bytes for a row above /
  // below the map), one vector offset per window COLUMN (pixel x, this lane's channel; a negative x gets an offset
  // beyond any row): the unfold's zero padding on all four sides then comes from the hardware's range check - no
  // per-position conditions, ~10 scalar instructions per window row instead of ~12 per window position.
  constexpr int EB = DT == FM_F32 ? 4 : 2;          // bytes per element
  constexpr int PXB = 64 * EB;                      // bytes per pixel (64 channels)
  static_assert(!(NCHW0 && DT != FM_F32), "half-precision maps are read channels-last");
  const char* base0 = reinterpret_cast<const char*>(map0) + (long)b * Hf0 * Wf0 * PXB;
  const char* base1 = reinterpret_cast<const char*>(map1) + (long)b * Hf1 * Wf1 * PXB;
  unsigned vo0[W], vo1[W];
#pragma unroll
  for (int wx = 0; wx < W; ++wx) {
    const int x0 = ox0 + wx, x1 = ox1 + wx;
    vo0[wx] = x0 >= 0 ? (unsigned)(x0 * PXB + lane * EB) : 0x80000000u;
    vo1[wx] = x1 >= 0 ? (unsigned)(x1 * PXB + lane * EB) : 0x80000000u;
  }
  auto load_elem = [](const __amdgpu_buffer_rsrc_t r, unsigned vo) -> float {
    if constexpr (DT == FM_F32) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0));
    else return half_bits_to_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, vo, 0, 0), DT);
  };
  float f0[WW], f1[WW];
#pragma unroll
  for (int wy = 0; wy < W; ++wy) {
    const int y0 = oy0 + wy, y1 = oy1 + wy;
    const bool ok0 = y0 >= 0 && y0 < Hf0, ok1 = y1 >= 0 && y1 < Hf1;          // wave-uniform
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(base1 + (long)(ok1 ? y1 : 0) * Wf1 * PXB), 0, ok1 ? Wf1 * PXB : 0, 0x00020000);
    if (!NCHW0) {
      const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(base0 + (long)(ok0 ? y0 : 0) * Wf0 * PXB), 0, ok0 ? Wf0 * PXB : 0, 0x00020000);
#pragma unroll
      for (int wx = 0; wx < W; ++wx) f0[wy * W + wx] = load_elem(r0, vo0[wx]);
    }
#pragma unroll
    for (int wx = 0; wx < W; ++wx) f1[wy * W + wx] = load_elem(r1, vo1[wx]);
  }
  if (NCHW0) {      // image 0 straight from the NCHW map: the crop's loader into a wave-private tile, read back by channel
    float* tile = tile_all + wv * kGatherTileFloats(W);
    wave_load_window64<W>(reinterpret_cast<const float*>(base0), Hf0, Wf0, oy0, ox0, tile, lane);
#pragma unroll
    for (int r = 0; r < WW; ++r) f0[r] = tile[r * 68 + lane];
  }
  fine_core<W>(f0, f1, lane, mix0, mix1, kc0[m * 2], kc0[m * 2 + 1], kc1[m * 2], kc1[m * 2 + 1], scale_f,
               out0 + (long)m * 3, out1 + (long)m * 3);
  }
}


// [N, 64, Hf, Wf] -> [N, Hf, Wf, 64]: one workgroup per (sample, row y, 64 pixels of the row); reads 64 channel
// segments of 256 bytes (16-byte loads along x), writes one contiguous 16 KiB block; the transpose goes through an
// LDS tile with an odd pitch.  grid (ceil(Wf / 64), Hf, N).
// T = float, or unsigned short for float16 / bfloat16 maps (2-byte elements moved as they are: half the bytes).
template <typename T>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc64(const T* __restrict__ src, T* __restrict__ dst, int Hf, int Wf,
                                                        int N) {
  __shared__ T tile[64 * 65];                  // [x][c], pitch 65
  nchw_to_nhwc64_units<T>(src, dst, Hf, Wf, N, tile, blockIdx.x, gridDim.x);     // (fm_maps_device.h)
}

}  // namespace fm

using namespace fm;

// ---- launch helpers shared by the plain and the merging entry points ----
template <bool MERGE>
static void launch_list64(int W, int blocks, hipStream_t st, const float* feat_f, int Hf, int Wf, int stride, int pad,
                          int w_c, const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max,
                          float* out, const half8* wpack, const float* ctx, int ctx_cells) {
  if (W == 5)
    hipLaunchKernelGGL((k_gather_nchw64<5, MERGE>), dim3(blocks), dim3(256), 0, st, feat_f, Hf, Wf, stride, pad, w_c,
                       b_ids, ids, d_count, m_max, out, wpack, ctx, ctx_cells);
  else
    hipLaunchKernelGGL((k_gather_nchw64<7, MERGE>), dim3(blocks), dim3(256), 0, st, feat_f, Hf, Wf, stride, pad, w_c,
                       b_ids, ids, d_count, m_max, out, wpack, ctx, ctx_cells);
}
template <bool MERGE>
static void launch_cells64(int W, int blocks, int nimg, hipStream_t st, const CellArgs& a) {
  if (W == 5) hipLaunchKernelGGL((k_gather_cellorder64<5, MERGE>), dim3(blocks, nimg), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_gather_cellorder64<7, MERGE>), dim3(blocks, nimg), dim3(256), 0, st, a);
}
static CellImage cell_image(const float* feat, int N, int Hf, int Wf, int h_c, int w_c, const int32_t* map, int pitch,
                            const int32_t* ties, const int64_t* ids, float* out, const float* ctx) {
  CellImage im;
  im.feat = feat; im.Hf = Hf; im.Wf = Wf; im.w_c = w_c; im.cells = h_c * w_c; im.total_cells = N * h_c * w_c;
  im.cell_to_match = map; im.cell_pitch = pitch; im.ties = ties; im.ids = ids; im.out = out; im.ctx = ctx;
  return im;
}
// the fast NCHW kernels address one sample's map with 31-bit byte offsets (buffer descriptor)
static bool fast_nchw64(int Cf, int Hf, int Wf, int W) {
  return Cf == 64 && (W == 5 || W == 7) && (long)Hf * Wf * 64 * 4 < (1L << 31);
}
// 8 XCD ranges of ceil(M/8) windows, four windows (waves) per workgroup
static int list_blocks(int m_max) { return 8 * (((m_max + 7) / 8 + 3) / 4); }
// largest grid of the kernels of fm_fine_match_maps (a multiple of 8; they walk longer lists in grid strides)
constexpr int kFineGridCap = 1 << 20;
// one wave per cell, four per workgroup, grid a multiple of 8
static int cell_blocks(long total) { return (int)(((total + 3) / 4 + 7) / 8 * 8); }

template <int DT>
static void launch_generic_gather(const void* feat_f, int Cf, int Hf, int Wf, int layout, int W, int stride, int pad, int w_c,
                                  const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max, float* out,
                                  hipStream_t st) {
  using E = typename MapElem<DT>::type;
  if (layout == 0) {
    const size_t smem = (size_t)W * W * (Cf + 1) * sizeof(float);
    hipLaunchKernelGGL(k_gather_nchw<DT>, dim3(m_max), dim3(256), smem, st, (const E*)feat_f, Cf, Hf, Wf, W, stride, pad, w_c,
                       b_ids, ids, d_count, m_max, out);
  } else {
    hipLaunchKernelGGL(k_gather_nhwc<DT>, dim3(m_max), dim3(256), 0, st, (const E*)feat_f, Cf, Hf, Wf, W, stride, pad, w_c,
                       b_ids, ids, d_count, m_max, out);
  }
}

extern "C" int fm_gather_windows_dtype(const void* feat_f, int map_dtype, int N, int Cf, int Hf, int Wf, int layout, int W,
                                       int stride, int pad, int w_c, const int64_t* b_ids, const int64_t* ids,
                                       const int32_t* d_count, int m_max, float* out, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f || !b_ids || !ids || !out) return FM_E_NULL;
  if (map_dtype != FM_F32 && map_dtype != FM_F16 && map_dtype != FM_BF16) return FM_E_UNSUPPORTED;
  if (N <= 0 || Cf <= 0 || Hf <= 0 || Wf <= 0 || W <= 0 || stride <= 0 || w_c <= 0 || m_max < 0) return FM_E_SHAPE;
  if (W > 15 || Cf > 512 || (layout == 1 && Cf % 4) || (layout != 0 && layout != 1)) return FM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0 && (size_t)W * W * (Cf + 1) * sizeof(float) > 64 * 1024 &&
      !(map_dtype == FM_F32 && fast_nchw64(Cf, Hf, Wf, W))) return FM_E_UNSUPPORTED;
  if (map_dtype == FM_F32 && layout == 0 && fast_nchw64(Cf, Hf, Wf, W)) {
    launch_list64<false>(W, list_blocks(m_max), st, (const float*)feat_f, Hf, Wf, stride, pad, w_c, b_ids, ids, d_count, m_max,
                         out, nullptr, nullptr, 0);
  } else if (map_dtype == FM_F32 && layout == 1 && Cf == 64 && (W == 5 || W == 7)) {
    // channels-last fast path: one wave per window, 16-byte chunks
    if (W == 5)
      hipLaunchKernelGGL(k_gather_nhwc64<5>, dim3(list_blocks(m_max)), dim3(256), 0, st, (const float*)feat_f, Hf, Wf, stride,
                         pad, w_c, b_ids, ids, d_count, m_max, out);
    else
      hipLaunchKernelGGL(k_gather_nhwc64<7>, dim3(list_blocks(m_max)), dim3(256), 0, st, (const float*)feat_f, Hf, Wf, stride,
                         pad, w_c, b_ids, ids, d_count, m_max, out);
  } else if (map_dtype == FM_F32) {
    launch_generic_gather<FM_F32>(feat_f, Cf, Hf, Wf, layout, W, stride, pad, w_c, b_ids, ids, d_count, m_max, out, st);
  } else if (map_dtype == FM_F16) {
    // half-precision maps (an autocast backbone's hand-over): read as they are - every value is exact in float32 -
    // by the generic kernels; no up-cast pass over the whole map in front of the crop
    launch_generic_gather<FM_F16>(feat_f, Cf, Hf, Wf, layout, W, stride, pad, w_c, b_ids, ids, d_count, m_max, out, st);
  } else {
    launch_generic_gather<FM_BF16>(feat_f, Cf, Hf, Wf, layout, W, stride, pad, w_c, b_ids, ids, d_count, m_max, out, st);
  }
  return (int)hipGetLastError();
}

extern "C" int fm_gather_windows(const float* feat_f, int N, int Cf, int Hf, int Wf, int layout, int W, int stride,
                                 int pad, int w_c, const int64_t* b_ids, const int64_t* ids, const int32_t* d_count,
                                 int m_max, float* out, void* stream) {
  return fm_gather_windows_dtype(feat_f, FM_F32, N, Cf, Hf, Wf, layout, W, stride, pad, w_c, b_ids, ids, d_count, m_max, out,
                                 stream);
}

extern "C" int fm_gather_windows_cells(const float* feat_f, int N, int Cf, int Hf, int Wf, int W, int stride, int pad,
                                       int h_c, int w_c, const int32_t* cell_to_match, int cell_pitch,
                                       const int32_t* ties, const int64_t* b_ids, const int64_t* ids,
                                       const int32_t* d_count, int m_max, float* out, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f || !cell_to_match || !ties || !b_ids || !ids || !out) return FM_E_NULL;
  if (N <= 0 || Hf <= 0 || Wf <= 0 || h_c <= 0 || w_c <= 0 || stride <= 0 || m_max < 0 || cell_pitch < h_c * w_c)
    return FM_E_SHAPE;
  if (!fast_nchw64(Cf, Hf, Wf, W)) return FM_E_UNSUPPORTED;
  CellArgs a;
  a.im[0] = a.im[1] = cell_image(feat_f, N, Hf, Wf, h_c, w_c, cell_to_match, cell_pitch, ties, ids, out, nullptr);
  a.stride = stride; a.pad = pad; a.m_max = m_max; a.b_ids = b_ids; a.d_count = d_count; a.wpack = nullptr;
  launch_cells64<false>(W, cell_blocks((long)N * h_c * w_c), 1, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// merge_feat.weight[:, :64] (row-major [64, 128], fine_preprocess.py:26) -> MFMA B fragments, f16 hi and lo:
// packed[((nt*4 + ks)*2 + plane)*64 + lane][j] = plane(W_w[32*nt + lane%32][16*ks + 8*(lane/32) + j])
__global__ __launch_bounds__(256) void k_merge_pack(const float* __restrict__ merge_w, half8* __restrict__ packed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;       // (nt, ks, lane): 2 * 4 * 64 = 512
  if (idx >= 512) return;
  const int lane = idx & 63, ks = (idx >> 6) & 3, nt = idx >> 8;
  const float* src = merge_w + (32 * nt + (lane & 31)) * 128 + 16 * ks + 8 * (lane >> 5);
  half8 hh, ll;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = src[j] * kMergeWgtScale;
    // a weight beyond the fixed scale (|w| >= 16) would turn into +-inf here and into a wrong finite number or NaN
    // somewhere downstream: make it NaN for certain - every output that touches it then says so
    if (!(fabsf(x) <= 65504.f)) x = __builtin_nanf("");
    hh[j] = (_Float16)x;
    ll[j] = (_Float16)(x - (float)hh[j]);
  }
  packed[((nt * 4 + ks) * 2 + 0) * 64 + lane] = hh;
  packed[((nt * 4 + ks) * 2 + 1) * 64 + lane] = ll;
}

extern "C" int fm_merge_pack_weights(const float* merge_w, int Cf, void* packed, void* stream) {
  if (!merge_w || !packed) return FM_E_NULL;
  if (Cf != 64) return FM_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_merge_pack, dim3(2), dim3(256), 0, (hipStream_t)stream, merge_w, (half8*)packed);
  return (int)hipGetLastError();
}

extern "C" int fm_gather_merge_windows(const float* feat_f, int N, int Cf, int Hf, int Wf, int W, int stride, int pad,
                                       int h_c, int w_c, const int32_t* cell_to_match, int cell_pitch,
                                       const int32_t* ties, const void* packed_w, const float* ctx_bias,
                                       const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max,
                                       float* out, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f || !packed_w || !ctx_bias || !b_ids || !ids || !out) return FM_E_NULL;
  if (cell_to_match && !ties) return FM_E_NULL;
  if (N <= 0 || Hf <= 0 || Wf <= 0 || h_c <= 0 || w_c <= 0 || stride <= 0 || m_max < 0) return FM_E_SHAPE;
  if (cell_to_match && cell_pitch < h_c * w_c) return FM_E_SHAPE;
  if (!fast_nchw64(Cf, Hf, Wf, W)) return FM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)N * h_c * w_c;
  if (cell_to_match) {
    CellArgs a;
    a.im[0] = a.im[1] = cell_image(feat_f, N, Hf, Wf, h_c, w_c, cell_to_match, cell_pitch, ties, ids, out, ctx_bias);
    a.stride = stride; a.pad = pad; a.m_max = m_max; a.b_ids = b_ids; a.d_count = d_count;
    a.wpack = (const half8*)packed_w;
    launch_cells64<true>(W, cell_blocks(total), 1, st, a);
  }
  else
    launch_list64<true>(W, list_blocks(m_max), st, feat_f, Hf, Wf, stride, pad, w_c, b_ids, ids, d_count, m_max, out,
                        (const half8*)packed_w, ctx_bias, h_c * w_c);
  return (int)hipGetLastError();
}

extern "C" int fm_gather_windows_pair(const float* feat_f0, const float* feat_f1, int N, int Cf, int Hf0, int Wf0,
                                      int Hf1, int Wf1, int W, int stride, int pad, int h0c, int w0c, int h1c, int w1c,
                                      const int32_t* cell0, int pitch0, const int32_t* ties0, const int32_t* cell1,
                                      int pitch1, const int32_t* ties1, const void* packed_w, const float* ctx0,
                                      const float* ctx1, const int64_t* b_ids, const int64_t* i_ids,
                                      const int64_t* j_ids, const int32_t* d_count, int m_max, float* out0,
                                      float* out1, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f0 || !feat_f1 || !cell0 || !cell1 || !ties0 || !ties1 || !b_ids || !i_ids || !j_ids || !out0 || !out1)
    return FM_E_NULL;
  if (packed_w && (!ctx0 || !ctx1)) return FM_E_NULL;
  if (N <= 0 || Hf0 <= 0 || Wf0 <= 0 || Hf1 <= 0 || Wf1 <= 0 || h0c <= 0 || w0c <= 0 || h1c <= 0 || w1c <= 0 ||
      stride <= 0 || m_max < 0 || pitch0 < h0c * w0c || pitch1 < h1c * w1c)
    return FM_E_SHAPE;
  if (!fast_nchw64(Cf, Hf0, Wf0, W) || !fast_nchw64(Cf, Hf1, Wf1, W)) return FM_E_UNSUPPORTED;
  CellArgs a;
  a.im[0] = cell_image(feat_f0, N, Hf0, Wf0, h0c, w0c, cell0, pitch0, ties0, i_ids, out0, ctx0);
  a.im[1] = cell_image(feat_f1, N, Hf1, Wf1, h1c, w1c, cell1, pitch1, ties1, j_ids, out1, ctx1);
  a.stride = stride; a.pad = pad; a.m_max = m_max; a.b_ids = b_ids; a.d_count = d_count;
  a.wpack = (const half8*)packed_w;
  const long t0 = (long)N * h0c * w0c, t1 = (long)N * h1c * w1c;
  const int blocks = cell_blocks(t0 > t1 ? t0 : t1);
  if (packed_w) launch_cells64<true>(W, blocks, 2, (hipStream_t)stream, a);
  else launch_cells64<false>(W, blocks, 2, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int fm_fine_match(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                             const float* mix0, const float* mix1, const float* mkpts0_c, const float* mkpts1_c,
                             float scale_f, float* out0, float* out1, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!win0 || !win1 || !mix0 || !mix1 || !mkpts0_c || !mkpts1_c || !out0 || !out1) return FM_E_NULL;
  if (m_max < 0) return FM_E_SHAPE;
  if (Cf != 64 || (WW != 25 && WW != 49)) return FM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (m_max + 3) / 4;
  if (WW == 49)
    hipLaunchKernelGGL(k_fine<7>, dim3(blocks), dim3(256), 0, st, win0, win1, m_max, d_count, mix0, mix1, mkpts0_c,
                       mkpts1_c, scale_f, out0, out1);
  else
    hipLaunchKernelGGL(k_fine<5>, dim3(blocks), dim3(256), 0, st, win0, win1, m_max, d_count, mix0, mix1, mkpts0_c,
                       mkpts1_c, scale_f, out0, out1);
  return (int)hipGetLastError();
}

// the fine maps of one sample must be addressable with 31-bit byte offsets (buffer descriptor)
static bool maps64_ok(int Cf, int Hf, int Wf, int W) {
  return Cf == 64 && (W == 5 || W == 7) && (long)Hf * Wf * 256 < (1L << 31);
}

extern "C" size_t fm_fine_maps_scratch_bytes_dtype(int N, int Cf, int Hf0, int Wf0, int Hf1, int Wf1, int layout, int map_dtype) {
  if (layout != 0 || N <= 0 || Cf <= 0 || Hf0 <= 0 || Wf0 <= 0 || Hf1 <= 0 || Wf1 <= 0) return 0;
  if (map_dtype == FM_F32) return (size_t)N * Cf * 4 * (size_t)Hf1 * Wf1;      // the channels-last copy of image 1
  // half-precision maps: channels-last copies of BOTH images (2-byte elements), image 0's first, 256-byte aligned
  const size_t b0 = (size_t)N * Cf * 2 * (size_t)Hf0 * Wf0, b1 = (size_t)N * Cf * 2 * (size_t)Hf1 * Wf1;
  return align256(b0) + b1;
}
extern "C" size_t fm_fine_maps_scratch_bytes(int N, int Cf, int Hf0, int Wf0, int Hf1, int Wf1, int layout) {
  return fm_fine_maps_scratch_bytes_dtype(N, Cf, Hf0, Wf0, Hf1, Wf1, layout, FM_F32);
}

extern "C" int fm_fine_match_maps_dtype(const void* feat_f0, const void* feat_f1, int map_dtype, int layout, int N, int Cf,
                                        int Hf0, int Wf0, int Hf1, int Wf1, int W, int stride, int pad, int w0c, int w1c,
                                        const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids,
                                        const int32_t* d_count, int m_max, const float* mix0, const float* mix1,
                                        const float* mkpts0_c, const float* mkpts1_c, float scale_f, void* scratch,
                                        float* out0, float* out1, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f0 || !feat_f1 || !b_ids || !i_ids || !j_ids || !mix0 || !mix1 || !mkpts0_c || !mkpts1_c || !out0 || !out1)
    return FM_E_NULL;
  if (N <= 0 || Hf0 <= 0 || Wf0 <= 0 || Hf1 <= 0 || Wf1 <= 0 || stride <= 0 || w0c <= 0 || w1c <= 0 || m_max < 0)
    return FM_E_SHAPE;
  if (map_dtype != FM_F32 && map_dtype != FM_F16 && map_dtype != FM_BF16) return FM_E_UNSUPPORTED;
  if ((layout != 0 && layout != 1 && layout != FM_LAYOUT_NCHW_PREPARED) || !maps64_ok(Cf, Hf0, Wf0, W) ||
      !maps64_ok(Cf, Hf1, Wf1, W))
    return FM_E_UNSUPPORTED;
  // (image 1's channels-last copy was made by the coarse call - fm_coarse_match_maps - into `scratch`: float32 maps only)
  const bool prepared = layout == FM_LAYOUT_NCHW_PREPARED;
  if (prepared && map_dtype != FM_F32) return FM_E_UNSUPPORTED;
  if (prepared) layout = 0;
  if (layout == 0 && !scratch) return FM_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const float* m0 = (const float*)feat_f0;
  const float* m1 = (const float*)feat_f1;
  int grid_cap = kFineGridCap;
#ifdef FM_TUNE_ENV
  if (const char* e = getenv("FM_FINE_GRID")) grid_cap = atoi(e) > 0 ? atoi(e) / 8 * 8 : kFineGridCap;
#endif
  const bool half = map_dtype != FM_F32;
  if (layout == 0 && !half) {   // NCHW: a channels-last copy of image 1 (coalesced on both sides); image 0 is read as it is
    float* s1 = (float*)scratch;
    const long pieces = (long)((Wf1 + 63) / 64) * Hf1 * N;
    if (!prepared)
      hipLaunchKernelGGL(k_nchw_to_nhwc64<float>, dim3((unsigned)(pieces < grid_cap ? pieces : grid_cap)), dim3(256), 0, st,
                         (const float*)feat_f1, s1, Hf1, Wf1, N);
    m1 = s1;
  } else if (layout == 0) {     // NCHW float16 / bfloat16: channels-last copies of both maps, element type kept
    unsigned short* s0 = (unsigned short*)scratch;
    unsigned short* s1 = (unsigned short*)((char*)scratch + align256((size_t)N * Cf * 2 * (size_t)Hf0 * Wf0));
    const long p0 = (long)((Wf0 + 63) / 64) * Hf0 * N, p1 = (long)((Wf1 + 63) / 64) * Hf1 * N;
    hipLaunchKernelGGL(k_nchw_to_nhwc64<unsigned short>, dim3((unsigned)(p0 < grid_cap ? p0 : grid_cap)), dim3(256), 0, st,
                       (const unsigned short*)feat_f0, s0, Hf0, Wf0, N);
    hipLaunchKernelGGL(k_nchw_to_nhwc64<unsigned short>, dim3((unsigned)(p1 < grid_cap ? p1 : grid_cap)), dim3(256), 0, st,
                       (const unsigned short*)feat_f1, s1, Hf1, Wf1, N);
    m0 = (const float*)s0;
    m1 = (const float*)s1;
  }
  const int blocks = list_blocks(m_max) < grid_cap ? list_blocks(m_max) : grid_cap;
#define FM_FINE_MAPS_LAUNCH(WQ, N0, DTQ)                                                                                 \
  hipLaunchKernelGGL((k_fine_maps<WQ, N0, DTQ>), dim3(blocks), dim3(256), 0, st, m0, m1, Hf0, Wf0, Hf1, Wf1, stride, pad, \
                     w0c, w1c, b_ids, i_ids, j_ids, d_count, m_max, mix0, mix1, mkpts0_c, mkpts1_c, scale_f, out0, out1)
#define FM_FINE_MAPS_W(WQ)                                                        \
  if (map_dtype == FM_F16) FM_FINE_MAPS_LAUNCH(WQ, false, FM_F16);                \
  else if (map_dtype == FM_BF16) FM_FINE_MAPS_LAUNCH(WQ, false, FM_BF16);         \
  else if (layout == 0) FM_FINE_MAPS_LAUNCH(WQ, true, FM_F32);                    \
  else FM_FINE_MAPS_LAUNCH(WQ, false, FM_F32)
  if (W == 5) { FM_FINE_MAPS_W(5); } else { FM_FINE_MAPS_W(7); }
#undef FM_FINE_MAPS_W
#undef FM_FINE_MAPS_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int fm_fine_match_maps(const float* feat_f0, const float* feat_f1, int layout, int N, int Cf, int Hf0, int Wf0,
                                  int Hf1, int Wf1, int W, int stride, int pad, int w0c, int w1c, const int64_t* b_ids,
                                  const int64_t* i_ids, const int64_t* j_ids, const int32_t* d_count, int m_max,
                                  const float* mix0, const float* mix1, const float* mkpts0_c, const float* mkpts1_c,
                                  float scale_f, void* scratch, float* out0, float* out1, void* stream) {
  return fm_fine_match_maps_dtype(feat_f0, feat_f1, FM_F32, layout, N, Cf, Hf0, Wf0, Hf1, Wf1, W, stride, pad, w0c, w1c,
                                  b_ids, i_ids, j_ids, d_count, m_max, mix0, mix1, mkpts0_c, mkpts1_c, scale_f, scratch,
                                  out0, out1, stream);
}
