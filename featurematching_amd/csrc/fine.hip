// Fine stage: window crop (network/module/fine_preprocess.py:43-50) and the dual-direction
// local-window correlation + soft-argmax (network/utils/fine_matching_new.py:50-79).
#include "fm_internal.h"

namespace fm {

// Workgroups are dealt round-robin over the 8 XCDs (private L2s).  Give each XCD a contiguous range
// of windows: matches are sorted by coarse cell, so neighbouring windows share cache lines of the
// fine map and one XCD then pulls only its part of the map over the fabric (speed only).
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int q = n >> 3, rem = n & 7, x = bid & 7, y = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + y;
}

// ----------------------------------------------------------------------------------------
// k_gather_nchw64<W>: one workgroup per window (Cf = 64, W compile-time).
// The reference unfolds EVERY coarse cell's window (60 MB per image at 640x480) and then selects M
// of them; here only the M selected windows are read, straight from the NCHW map.  Lanes run along
// (channel, window row, x) with x fastest, so one wave-load touches ~64/W short runs in 2-3 channel
// planes.  (Lane = channel would put the 64 lanes of a load one plane stride apart - 307200 B at
// 240x320 - which lands on a handful of L2/HBM channels; measured 2x slower.)  The [c][r] -> [r][c]
// transpose goes through LDS (pitch 65: conflict-free both ways) and the window leaves as one
// contiguous record.  Window origin = stride*cell - pad with the reference's literal pad = 2, zero
// outside the map.
// ----------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_gather_nchw64(const float* __restrict__ feat, int Hf, int Wf, int stride,
                                                       int pad, int w_c, const int64_t* __restrict__ b_ids,
                                                       const int64_t* __restrict__ ids,
                                                       const int32_t* __restrict__ d_count, int m_max,
                                                       float* __restrict__ out) {
  constexpr int CF = 64, WW = W * W, TOTAL = CF * WW;
  __shared__ float tile[WW * (CF + 1)];
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  // spread the M live windows (not the m_max launched ones) contiguously over the XCDs
  const int per = (M + 7) >> 3;
  const int m = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || m >= M) return;
  const int b = (int)b_ids[m];
  const int id = (int)ids[m];
  const int cy = id / w_c;
  const int oy = cy * stride - pad;
  const int ox = (id - cy * w_c) * stride - pad;
  const float* src = feat + (long)b * CF * Hf * Wf;
#pragma unroll
  for (int it = 0; it < (TOTAL + 255) / 256; ++it) {
    const int idx = it * 256 + threadIdx.x;
    if (idx < TOTAL) {
      const int c = idx / WW;
      const int rem = idx - c * WW;
      const int wy = rem / W, wx = rem - wy * W;
      const int y = oy + wy, x = ox + wx;
      float v = 0.f;
      if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = src[((long)c * Hf + y) * Wf + x];
      tile[rem * (CF + 1) + c] = v;
    }
  }
  __syncthreads();
  float* dst = out + (long)m * TOTAL;
#pragma unroll
  for (int it = 0; it < (TOTAL + 255) / 256; ++it) {
    const int idx = it * 256 + threadIdx.x;
    if (idx < TOTAL) dst[idx] = tile[(idx >> 6) * (CF + 1) + (idx & 63)];
  }
}

// ----------------------------------------------------------------------------------------
// k_gather_cells64<W>: cell-tiled crop.  One workgroup = 8 adjacent coarse cells of one coarse row.
// The strip of the fine map under those 8 windows (W rows x 28+W columns x 64 channels) is read ONCE
// in runs of ~34 floats (two cache lines per (channel,row) instead of one line per 20-byte run and
// window: 3x less fabric traffic than the per-window kernel), transposed through LDS, and the windows
// of the cells that are matched (cell -> match map from the coarse stage, 0 = unmatched) are written
// as contiguous records.  Matches that lost their cell to an exactly tied match are picked up at the
// end from this workgroup's slice of the match list with the per-window path.
// ----------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_gather_cells64(const float* __restrict__ feat, int N, int Hf, int Wf, int h_c,
                                                        int w_c, const int32_t* __restrict__ cell_to_match,
                                                        int cell_pitch, const int64_t* __restrict__ b_ids,
                                                        const int64_t* __restrict__ ids,
                                                        const int32_t* __restrict__ d_count, int m_max,
                                                        float* __restrict__ out) {
  constexpr int CF = 64, G = 8, STRIDE = 4, PAD = 2, WW = W * W, TOTAL = CF * WW;
  constexpr int SPAN = (G - 1) * STRIDE + W;        // fine columns under the 8 windows
  constexpr int NX2 = (SPAN + 1) / 2;               // loaded as float2 (the strip starts at an even column)
  constexpr int SPANP = 2 * NX2;
  constexpr int PITCH = CF + 1;
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [W][SPANP][PITCH]
  __shared__ int mids[G];
  __shared__ int left[32];
  __shared__ int nleft;
  const int tid = threadIdx.x;
  const int M = d_count ? min(d_count[0], m_max) : m_max;
  const int groups = (w_c + G - 1) / G;
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int gx = bid % groups;
  const int cy = (bid / groups) % h_c;
  const int b = bid / (groups * h_c);
  if (tid < G) {
    const int x = gx * G + tid;
    int m = -1;
    if (x < w_c) m = cell_to_match[(long)b * cell_pitch + cy * w_c + x] - 1;
    mids[tid] = (m < M) ? m : -1;
  }
  if (tid == 0) nleft = 0;
  __syncthreads();
  bool any = false;
#pragma unroll
  for (int k = 0; k < G; ++k) any = any || (mids[k] >= 0);
  if (any) {
    const int oy = cy * STRIDE - PAD, ox = gx * G * STRIDE - PAD;
    const float* src = feat + (long)b * CF * Hf * Wf;
    const bool even = (Wf & 1) == 0;
    // all loads of a thread are issued before the first LDS write (compile-time trip count, fully
    // unrolled): one round trip to L2/HBM per workgroup instead of one per loop iteration
    constexpr int NLOAD = (CF * W * NX2 + 255) / 256;
    float2 v[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = it * 256 + tid;
      const int xq = idx % NX2;
      const int cw = idx / NX2;
      const int wy = cw % W, c = cw / W;
      const int y = oy + wy, x = ox + 2 * xq;
      v[it] = make_float2(0.f, 0.f);
      if (idx < CF * W * NX2 && y >= 0 && y < Hf) {
        const float* row = src + ((long)c * Hf + y) * Wf;
        if (even && x >= 0 && x + 1 < Wf) v[it] = *reinterpret_cast<const float2*>(row + x);
        else {
          if (x >= 0 && x < Wf) v[it].x = row[x];
          if (x + 1 >= 0 && x + 1 < Wf) v[it].y = row[x + 1];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = it * 256 + tid;
      if (idx < CF * W * NX2) {
        const int xq = idx % NX2;
        const int cw = idx / NX2;
        const int wy = cw % W, c = cw / W;
        tile[(wy * SPANP + 2 * xq) * PITCH + c] = v[it].x;
        tile[(wy * SPANP + 2 * xq + 1) * PITCH + c] = v[it].y;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < G; ++k) {
      const int m = mids[k];
      if (m < 0) continue;
      float* dst = out + (long)m * TOTAL;
#pragma unroll
      for (int it = 0; it < (TOTAL + 255) / 256; ++it) {
        const int idx = it * 256 + tid;
        if (idx < TOTAL) {
          const int rpos = idx >> 6, c = idx & 63;
          const int wy = rpos / W, wx = rpos - wy * W;
          dst[idx] = tile[(wy * SPANP + k * STRIDE + wx) * PITCH + c];
        }
      }
    }
  }
  // ---- left-overs of this workgroup's slice of the match list (exact ties only) ----
  const int chunk = (M + (int)gridDim.x - 1) / (int)gridDim.x;
  const int m0 = blockIdx.x * chunk;
  if (tid < chunk && tid < 32 && m0 + tid < M) {
    const int m = m0 + tid;
    const int mb = (int)b_ids[m], id = (int)ids[m];
    if (cell_to_match[(long)mb * cell_pitch + id] != m + 1) left[atomicAdd(&nleft, 1)] = m;
  }
  __syncthreads();
  const int nl = nleft;
  for (int q = 0; q < nl; ++q) {          // rare
    __syncthreads();
    const int m = left[q];
    const int mb = (int)b_ids[m], id = (int)ids[m];
    const int ccy = id / w_c;
    const int oy = ccy * STRIDE - PAD, ox = (id - ccy * w_c) * STRIDE - PAD;
    const float* src = feat + (long)mb * CF * Hf * Wf;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int c = idx / WW, rem = idx - c * WW;
      const int wy = rem / W, wx = rem - wy * W;
      const int y = oy + wy, x = ox + wx;
      float v = 0.f;
      if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = src[((long)c * Hf + y) * Wf + x];
      tile[rem * PITCH + c] = v;
    }
    __syncthreads();
    float* dst = out + (long)m * TOTAL;
    for (int idx = tid; idx < TOTAL; idx += 256) dst[idx] = tile[(idx >> 6) * PITCH + (idx & 63)];
  }
}

// generic fallback (any Cf / W): one workgroup per window, transposed through LDS
__global__ __launch_bounds__(256) void k_gather_nchw(const float* __restrict__ feat, int Cf, int Hf, int Wf, int W,
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
  const float* src = feat + (long)b * Cf * Hf * Wf;
  const int total = Cf * WW;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int c = idx / WW;
    const int rem = idx - c * WW;
    const int wy = rem / W, wx = rem - wy * W;
    const int y = oy + wy, x = ox + wx;
    float v = 0.f;
    if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = src[((long)c * Hf + y) * Wf + x];
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

// NHWC storage: every window row is W*Cf contiguous floats - a straight float4 copy.
__global__ __launch_bounds__(256) void k_gather_nhwc(const float* __restrict__ feat, int Cf, int Hf, int Wf, int W,
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
  const float4* src = reinterpret_cast<const float4*>(feat + (long)b * Hf * Wf * Cf);
  float4* dst = reinterpret_cast<float4*>(out + (long)m * WW * Cf);
  for (int idx = threadIdx.x; idx < WW * c4; idx += 256) {
    const int rpos = idx / c4;
    const int c = idx - rpos * c4;
    const int wy = rpos / W, wx = rpos - wy * W;
    const int y = oy + wy, x = ox + wx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = src[((long)y * Wf + x) * c4 + c];
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

// One butterfly step of the transpose-reduce: p[0..2*HALF) -> p[0..HALF); lanes with bit MASK set
// continue with the upper half of the sums, the others with the lower half.
template <int HALF, int MASK, int NP>
__device__ __forceinline__ void tr_step(float (&p)[NP], int lane) {
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const float a = pair_op<MASK, OpAdd>(p[k]);
    const float b = pair_op<MASK, OpAdd>(p[k + HALF]);
    p[k] = up ? b : a;
  }
}

// in: p[k] = this lane's term of sum k.  out: the lane whose tr_index is k returns sum_lanes p[k].
// The steps with many pairs use the cheapest exchanges (quad permutes); see tr_index for the map.
template <int NP>
__device__ __forceinline__ float transpose_reduce(float (&p)[NP], int lane) {
  if constexpr (NP == 64) {
    tr_step<32, 1>(p, lane); tr_step<16, 2>(p, lane); tr_step<8, 8>(p, lane);
    tr_step<4, 4>(p, lane); tr_step<2, 16>(p, lane); tr_step<1, 32>(p, lane);
  } else {
    tr_step<16, 1>(p, lane); tr_step<8, 2>(p, lane); tr_step<4, 8>(p, lane);
    tr_step<2, 4>(p, lane); tr_step<1, 16>(p, lane);
    p[0] = pair_op<32, OpAdd>(p[0]);     // the two half-waves hold the two halves of the channels
  }
  return p[0];
}
template <int NP>
__device__ __forceinline__ int tr_index(int lane) {
  if constexpr (NP == 64)
    return 32 * (lane & 1) + 16 * ((lane >> 1) & 1) + 8 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1) +
           2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);
  else
    return 16 * (lane & 1) + 8 * ((lane >> 1) & 1) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 4) & 1);
}

// pos = window position this lane holds (sim is that position's similarity), on = lane takes part
template <int W>
__device__ __forceinline__ void soft_argmax(float sim, int pos, bool on, int lane, float inv_sqrt_c, float scale_f,
                                            float kx, float ky, float* out) {
  const float x = on ? sim * inv_sqrt_c : -INFINITY;
  const float mx = wave_max(x);
  const float e = on ? __expf(x - mx) : 0.f;
  const float heat = e / wave_sum(e);
  const int wy = pos / W, wx = pos - wy * W;
  const float gx = ((float)wx / (float)(W - 1) - 0.5f) * 2.f;     // kornia create_meshgrid, normalised
  const float gy = ((float)wy / (float)(W - 1) - 0.5f) * 2.f;
  const float cx = wave_sum(gx * heat), cy = wave_sum(gy * heat);
  const float vx = wave_sum(gx * gx * heat) - cx * cx;
  const float vy = wave_sum(gy * gy * heat) - cy * cy;
  if (lane == 0) {
    const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
    out[0] = kx + (cx * (float)(W / 2) * scale_f + (float)(W / 2));
    out[1] = ky + (cy * (float)(W / 2) * scale_f + (float)(W / 2));
    out[2] = sd;
  }
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
#pragma unroll
  for (int r = 0; r < WW; ++r) { f0[r] = p0[r * CF]; f1[r] = p1[r * CF]; }
  float q0 = mix0[WW], q1 = mix1[WW];
#pragma unroll
  for (int r = 0; r < WW; ++r) { q0 = __builtin_fmaf(mix0[r], f0[r], q0); q1 = __builtin_fmaf(mix1[r], f1[r], q1); }

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
  const bool on = pos < WW && lane < NP;
  soft_argmax<W>(sim0, pos, on, lane, inv_sqrt_c, scale_f, kc0[m * 2], kc0[m * 2 + 1], out0 + (long)m * 3);
  soft_argmax<W>(sim1, pos, on, lane, inv_sqrt_c, scale_f, kc1[m * 2], kc1[m * 2 + 1], out1 + (long)m * 3);
}

}  // namespace fm

using namespace fm;

extern "C" int fm_gather_windows(const float* feat_f, int N, int Cf, int Hf, int Wf, int layout, int W, int stride,
                                 int pad, int w_c, const int64_t* b_ids, const int64_t* ids, const int32_t* d_count,
                                 int m_max, float* out, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f || !b_ids || !ids || !out) return FM_E_NULL;
  if (N <= 0 || Cf <= 0 || Hf <= 0 || Wf <= 0 || W <= 0 || stride <= 0 || w_c <= 0 || m_max < 0) return FM_E_SHAPE;
  if (W > 15 || Cf > 512 || (layout == 1 && Cf % 4) || (layout != 0 && layout != 1)) return FM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0 && Cf == 64 && (W == 5 || W == 7)) {
    if (W == 5)
      hipLaunchKernelGGL(k_gather_nchw64<5>, dim3((m_max + 7) / 8 * 8), dim3(256), 0, st, feat_f, Hf, Wf, stride, pad, w_c, b_ids,
                         ids, d_count, m_max, out);
    else
      hipLaunchKernelGGL(k_gather_nchw64<7>, dim3((m_max + 7) / 8 * 8), dim3(256), 0, st, feat_f, Hf, Wf, stride, pad, w_c, b_ids,
                         ids, d_count, m_max, out);
  } else if (layout == 0) {
    const size_t smem = (size_t)W * W * (Cf + 1) * sizeof(float);
    if (smem > 64 * 1024) return FM_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_gather_nchw, dim3(m_max), dim3(256), smem, st, feat_f, Cf, Hf, Wf, W, stride, pad, w_c,
                       b_ids, ids, d_count, m_max, out);
  } else {
    hipLaunchKernelGGL(k_gather_nhwc, dim3(m_max), dim3(256), 0, st, feat_f, Cf, Hf, Wf, W, stride, pad, w_c, b_ids,
                       ids, d_count, m_max, out);
  }
  return (int)hipGetLastError();
}

extern "C" int fm_gather_windows_cells(const float* feat_f, int N, int Cf, int Hf, int Wf, int W, int stride, int pad,
                                       int h_c, int w_c, const int32_t* cell_to_match, int cell_pitch,
                                       const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max,
                                       float* out, void* stream) {
  if (m_max == 0) return FM_OK;
  if (!feat_f || !cell_to_match || !b_ids || !ids || !out) return FM_E_NULL;
  if (N <= 0 || Hf <= 0 || Wf <= 0 || h_c <= 0 || w_c <= 0 || m_max < 0 || cell_pitch < h_c * w_c) return FM_E_SHAPE;
  if (Cf != 64 || (W != 5 && W != 7) || stride != 4 || pad != 2) return FM_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int groups = (w_c + 7) / 8;
  const int blocks = N * h_c * groups;
  if (m_max > 32L * blocks) return FM_E_UNSUPPORTED;      // left-over slices hold at most 32 matches each
  const int span = 7 * 4 + W;
  const size_t smem = (size_t)W * (2 * ((span + 1) / 2)) * 65 * sizeof(float);
  hipError_t e;
  if (W == 5) {
    static unsigned long long lds_set5 = 0;
    e = ensure_dynamic_lds(&k_gather_cells64<5>, (int)smem, &lds_set5);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_gather_cells64<5>, dim3(blocks), dim3(256), smem, st, feat_f, N, Hf, Wf, h_c, w_c, cell_to_match,
                       cell_pitch, b_ids, ids, d_count, m_max, out);
  } else {
    static unsigned long long lds_set7 = 0;
    e = ensure_dynamic_lds(&k_gather_cells64<7>, (int)smem, &lds_set7);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_gather_cells64<7>, dim3(blocks), dim3(256), smem, st, feat_f, N, Hf, Wf, h_c, w_c, cell_to_match,
                       cell_pitch, b_ids, ids, d_count, m_max, out);
  }
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
