// What does the SHAPE of a store stream cost on the way to HBM?  The dense conf_matrix sweep (k_dense<C, kDenseConf>)
// writes 5.9 GB at cfg#3 as tiles: a wave owns 32 rows and walks along them 32 columns (128 B per row) at a time, one
// store instruction = 2 rows x 128 B; a plain fill of the same bytes runs at 6.9 TB/s, the sweep's stores at ~3.
// Variants of a store-only kernel on the [64, 4800, 4800] float32 matrix:
//   fill        : contiguous (grid-stride float4)
//   tile W R    : workgroup = 8 waves; the waves are arranged RW x CW (RW * CW = 8): a wave owns 32 rows and, per step,
//                 W columns (W = 32: 2 rows x 128 B per instruction as in the sweep; 64: one row x 256 B per instruction);
//                 the workgroup's CW waves take adjacent column chunks: a contiguous run of CW * W * 4 bytes per row and step
//   nt          : the same with nontemporal stores
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_store_pattern.hip -o build/mbstore && ./build/mbstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_fill(float4* p, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// grid: (splits, row groups of RW * 32 rows, N); the workgroup walks its column range [c0, c1) in steps of CW * W columns
template <int W, int CW, bool NT>
__global__ __launch_bounds__(512) void k_tile(float* conf, int L, int S, int splits, int delay) {
  constexpr int RW = 8 / CW;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wv / CW, wc = wv % CW;
  const int row0 = (blockIdx.y * RW + wr) * 32;
  if (row0 >= L) return;
  const int per = ((S + splits - 1) / splits + CW * W - 1) / (CW * W) * (CW * W);
  const int c0 = blockIdx.x * per, c1 = min(S, c0 + per);
  float* base = conf + ((long)blockIdx.z * L + row0) * S;
  constexpr int RPI = 64 / W;                          // rows per store instruction
  const int lr = lane / W, lc = lane % W;
  for (int c = c0 + wc * W; c < c1; c += CW * W) {
    const float v = (float)c;
#pragma unroll 4
    for (int r = 0; r < 32; r += RPI) {
      const int row = row0 + r + lr, col = c + lc;
      if (row < L && col < S) {
        float* dst = base + (long)(r + lr) * S + col;
        if (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
    for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(8);
  }
}

// the same walk with 16-byte stores: 8 lanes cover one row's 32 columns, one instruction = 8 rows x 128 B (what an
// LDS transpose of the MFMA accumulators would allow)
__global__ __launch_bounds__(512) void k_tile_x4(float* conf, int L, int S, int splits) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.y * 8 + wv) * 32;
  if (row0 >= L) return;
  const int per = ((S + splits - 1) / splits + 31) / 32 * 32;
  const int c0 = blockIdx.x * per, c1 = min(S, c0 + per);
  float* base = conf + ((long)blockIdx.z * L + row0) * S;
  const int lr = lane >> 3, lc = (lane & 7) * 4;
  for (int c = c0; c < c1; c += 32) {
    const float v = (float)c;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
      if (row0 + r + lr < L) *reinterpret_cast<float4*>(base + (long)(r + lr) * S + c + lc) = make_float4(v, v, v, v);
    }
  }
}

// the sweep's whole memory stream: per unit the workgroup also READS 32 KiB of its sample's planes (4.9 MB per sample,
// shared by the sample's 19 workgroups, which sit on one XCD: L2 hits) - 16 bytes per lane, 4 loads per wave
__global__ __launch_bounds__(512) void k_tile_rw(float* conf, const float4* __restrict__ planes, int L, int S, float* sink, int lds_dma, int src, int nq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // all workgroups of a sample on one XCD: workgroup k of the grid runs on XCD k % 8
  const int per_s = (L + 255) / 256, nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lin = src == 2 ? (int)blockIdx.x : xcd * (nblk / 8) + idx;                // (nblk is a multiple of 8 here: 64 samples x 19)
  const int b = lin / per_s, panel = lin - b * per_s;
  const int row0 = (panel * 8 + wv) * 32;
  float* base = conf + ((long)b * L + row0) * S;
  // src 0: the sample's planes (shared by its 19 workgroups); 1: the same 32 KiB for everybody, every unit (always an L2
  // hit); 2: the sample's planes, but the workgroups of a sample spread over all XCDs (lin = blockIdx.x)
  const float4* pl = planes + (long)b * (S / 32) * 2048;          // 32 KiB per unit = 2048 float4
  const long ustride = src == 1 ? 0 : 2048;
  if (src == 1) pl = planes;
  const int lr = lane >> 5, lc = lane & 31;
  float acc = 0.f;
  for (int u = 0; u < S / 32; ++u) {
    if (lds_dma) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nq) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl + (long)u * ustride + (wv * 4 + q) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(smem + ((u & 3) * 32768) + (wv * 4 + q) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q < nq) { const float4 v = pl[(long)u * ustride + (wv * 4 + q) * 64 + lane]; acc += v.x + v.w; }
    }
    if (row0 < L) {
      const float v = (float)u + acc;
#pragma unroll 4
      for (int r = 0; r < 32; r += 2)
        if (row0 + r + lr < L) base[(long)(r + lr) * S + u * 32 + lc] = v;
    }
  }
  if (acc == 12345.f) *sink = acc;
}

template <int W, int CW, bool NT>
static void run(const char* name, float* conf, int N, int L, int S, int splits, int delay, hipEvent_t e0, hipEvent_t e1) {
  constexpr int RW = 8 / CW;
  const dim3 grid(splits, (L + RW * 32 - 1) / (RW * 32), N);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_tile<W, CW, NT>), grid, dim3(512), 0, 0, conf, L, S, splits, delay);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-28s W=%3d waves %dx%d splits %2d delay %d: %.3f ms  %.2f TB/s\n", name, W, RW, CW, splits, delay, best,
         4.0 * N * L * S / best / 1e9);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, L = 4800, S = 4800;
  const long n = (long)N * L * S;
  float* conf; if (hipMalloc(&conf, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, (float4*)conf, n / 4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("fill (grid-stride float4)                                          : %.3f ms  %.2f TB/s\n", ms, 4.0 * n / ms / 1e9);
  }
  for (int splits : {1, 3}) {
    run<32, 1, false>("tile (the sweep's shape)", conf, N, L, S, splits, 0, e0, e1);
    run<32, 1, true>("tile nt", conf, N, L, S, splits, 0, e0, e1);
    run<32, 2, false>("tile 2 chunks/row", conf, N, L, S, splits, 0, e0, e1);
    run<32, 4, false>("tile 4 chunks/row", conf, N, L, S, splits, 0, e0, e1);
    run<32, 8, false>("tile 8 chunks/row", conf, N, L, S, splits, 0, e0, e1);
    run<64, 1, false>("row x 256 B per instr", conf, N, L, S, splits, 0, e0, e1);
    run<64, 1, true>("row x 256 B per instr nt", conf, N, L, S, splits, 0, e0, e1);
    run<64, 4, false>("row x 256 B, 4 chunks/row", conf, N, L, S, splits, 0, e0, e1);
  }
  for (int splits : {1, 3}) {
    const dim3 grid(splits, (L + 255) / 256, N);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_tile_x4, grid, dim3(512), 0, 0, conf, L, S, splits);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("tile, 16-byte stores (8 rows x 128 B per instr) splits %d: %.3f ms  %.2f TB/s\n", splits, best, 4.0 * n / best / 1e9);
  }
  // one workgroup per compute unit at most (N = 4: 76 workgroups): the store path of ONE compute unit, HBM far from its limit
  for (int x4 = 0; x4 < 2; ++x4) {
    const dim3 grid(1, (L + 255) / 256, 4);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      if (x4) hipLaunchKernelGGL(k_tile_x4, grid, dim3(512), 0, 0, conf, L, S, 1);
      else hipLaunchKernelGGL((k_tile<32, 1, false>), grid, dim3(512), 0, 0, conf, L, S, 1, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("76 workgroups (8 waves each), %s stores: %.3f ms = %.1f GB/s per workgroup\n", x4 ? "16-byte" : "4-byte", best,
           4.0 * 256 * S * 4 / 4 / best / 1e6);
  }
  // the sweep's occupancy: ONE workgroup of 8 waves per compute unit (128 KiB of dynamic LDS requested, never touched)
  for (int delay : {0, 1, 2, 4}) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile<32, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const dim3 grid(1, (L + 255) / 256, N);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k_tile<32, 1, false>), grid, dim3(512), 128 * 1024, 0, conf, L, S, 1, delay);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("one workgroup per compute unit, delay %d: %.3f ms  %.2f TB/s\n", delay, best, 4.0 * n / best / 1e9);
  }
  {
    float4* planes; hipMalloc(&planes, (size_t)N * (S / 32) * 32768); hipMemset(planes, 0, (size_t)N * (S / 32) * 32768);
    float* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_rw), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int src = 0; src < 3; ++src)
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_tile_rw, dim3(N * ((L + 255) / 256)), dim3(512), 128 * 1024, 0, conf, planes, L, S, sink, mode, src, 4);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("stores + 32 KiB of plane reads per unit (%s; %s), one workgroup per compute unit: %.3f ms  %.2f TB/s of stores\n",
             mode ? "LDS-DMA" : "register loads", src == 0 ? "sample's planes, sample on one XCD" : (src == 1 ? "the same 32 KiB always" : "sample's planes, sample spread over XCDs"),
             best, 4.0 * n / best / 1e9);
    }
    for (int nq : {2, 1}) {                      // fewer plane bytes per unit: 16 KiB (one plane), 8 KiB (one plane, 512-row workgroups)
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_tile_rw, dim3(N * ((L + 255) / 256)), dim3(512), 128 * 1024, 0, conf, planes, L, S, sink, 1, 0, nq);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("stores + %d KiB of plane reads per unit (LDS-DMA; sample on one XCD): %.3f ms  %.2f TB/s of stores\n", 8 * nq, best, 4.0 * n / best / 1e9);
    }
    hipFree(planes); hipFree(sink);
  }
  // paced like the sweep (a unit of the real kernel takes ~1.5 us per wave): does the shape matter less when the stores trickle?
  run<32, 1, false>("tile, paced", conf, N, L, S, 3, 4, e0, e1);
  run<32, 1, true>("tile nt, paced", conf, N, L, S, 3, 4, e0, e1);
  run<32, 8, false>("tile 8 chunks/row, paced", conf, N, L, S, 3, 4, e0, e1);
  run<64, 1, false>("row x 256 B, paced", conf, N, L, S, 3, 4, e0, e1);
  hipFree(conf);
  return 0;
}
