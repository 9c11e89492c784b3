// Tiled NCHW -> channels-last transpose of 64-channel maps, shared by k_nchw_to_nhwc64 (fine.hip) and by the side-job role
// of the assignment kernel (coarse_select.hip: fm_coarse_match_maps).
#pragma once
#include <hip/hip_runtime.h>

namespace fm {

// dst [N, Hf, Wf, 64] <- src [N, 64, Hf, Wf]: (sample, row, 64-pixel piece) units in a flat order, this workgroup (256
// threads) takes units first, first + step, ...  Both sides coalesced: a unit is read as 64 channel rows of 64 pixels
// (16-byte loads) into an LDS tile with an odd pitch and written as 64 pixels of 64 channels (16-byte stores).
// T = float, or unsigned short for float16 / bfloat16 maps (2-byte elements moved as they are).  tile: 64 * 65 elements.
template <typename T>
__device__ __forceinline__ void nchw_to_nhwc64_units(const T* __restrict__ src, T* __restrict__ dst, int Hf, int Wf, int N,
                                                     T* tile, long first, long step) {
  struct alignas(4 * sizeof(T)) V4 { T x, y, z, w; };
  const int tid = threadIdx.x;
  const int tx = (Wf + 63) / 64;
  const long total = (long)tx * Hf * N;
  for (long t = first; t < total; t += step) {
    const int x0 = (int)(t % tx) * 64, y = (int)((t / tx) % Hf), b = (int)(t / ((long)tx * Hf));
    const T* in = src + ((long)b * 64 * Hf + y) * Wf + x0;      // + c * Hf * Wf
    const long plane = (long)Hf * Wf;
    const int nx = min(64, Wf - x0);
    const bool vec = (Wf & 3) == 0;              // rows aligned to 4 elements
    V4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {                // channel c = 16 p + tid / 16, pixels 4 (tid % 16) .. + 3
      const int c = 16 * p + (tid >> 4), xq = (tid & 15) * 4;
      const T* row = in + c * plane;
      if (vec && xq + 3 < nx) v[p] = *reinterpret_cast<const V4*>(row + xq);
      else v[p] = V4{xq < nx ? row[xq] : T(0), xq + 1 < nx ? row[xq + 1] : T(0), xq + 2 < nx ? row[xq + 2] : T(0),
                     xq + 3 < nx ? row[xq + 3] : T(0)};
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = 16 * p + (tid >> 4), xq = (tid & 15) * 4;
      tile[(xq + 0) * 65 + c] = v[p].x; tile[(xq + 1) * 65 + c] = v[p].y;
      tile[(xq + 2) * 65 + c] = v[p].z; tile[(xq + 3) * 65 + c] = v[p].w;
    }
    __syncthreads();
    V4* out = reinterpret_cast<V4*>(dst + (((long)b * Hf + y) * Wf + x0) * 64);
#pragma unroll
    for (int p = 0; p < 4; ++p) {                // pixel x = 16 p + tid / 16, channels 4 (tid % 16) .. + 3
      const int x = 16 * p + (tid >> 4), c4 = (tid & 15) * 4;
      if (x < nx)
        out[x * 16 + (tid & 15)] = V4{tile[x * 65 + c4], tile[x * 65 + c4 + 1], tile[x * 65 + c4 + 2], tile[x * 65 + c4 + 3]};
    }
    __syncthreads();                             // the tile is rewritten by the next unit
  }
}

}  // namespace fm
