// What ONE wave (and two waves per SIMD) can hide under int8 MFMAs on gfx950: cycles per v_mfma_i32_32x32x32_i8 /
// v_mfma_i32_16x16x64_i8 with F integer-maximum fillers (volatile asm, independent registers) in every gap, with and
// without a ds_read_b128 per two MFMAs.  Shapes the max pass's epilogue budget.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_i8_gap.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void vmax_i(int& d, int s) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(d) : "v"(s)); }
__device__ __forceinline__ int vmax3_i(int x, int y, int z) {
  int d;
  asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}

// SHAPE 0: 32x32x32 (16 acc regs), 1: 16x16x64 (4 acc regs).  CHAINS independent accumulators, round robin.
// FILL fillers behind every MFMA.  LDS: one ds_read_b128 per LDSP MFMAs (0 = none), read-ahead 4.
template <int SHAPE, int CHAINS, int FILL, int LDSP>
__global__ __launch_bounds__(512) void k(int* out, unsigned long long* stamps, int iters, int seed) {
  __shared__ __attribute__((aligned(16))) char smem[16384];
  v4i a[4], b;
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 4; ++i) a[q][i] = seed * (threadIdx.x % 7 + i + q) * 0x01010101;
  for (int i = 0; i < 4; ++i) b[i] = seed * (threadIdx.x % 5 + i) * 0x00010203;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((int*)smem)[i] = i * seed;
  __syncthreads();
  v16i c[CHAINS];
  v4i d[CHAINS];
  for (int q = 0; q < CHAINS; ++q) { for (int g = 0; g < 16; ++g) c[q][g] = 0; for (int g = 0; g < 4; ++g) d[q][g] = 0; }
  int f[16];
  for (int g = 0; g < 16; ++g) f[g] = seed + g;
  int src[8];
  for (int g = 0; g < 8; ++g) src[g] = seed * g + threadIdx.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
  v4i bq[4] = {b, b, b, b};
  v4i bw[2][8];
  for (int q = 0; q < 8; ++q) bw[0][q] = bw[1][q] = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      if (LDSP < 0) {       // burst: the 8 fragments of the NEXT 16-MFMA block at the head of this one
        if (j % 16 == 0) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bw[((j / 16) + 1) & 1][q]) : "v"(lds0), "n"(q * 1024));
        }
        if (j % 16 == 14) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      } else if (LDSP && j % LDSP == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[(j / LDSP) % 4]) : "v"(lds0), "n"((j % 8) * 1024));
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[(j / LDSP + 1) % 4]));
      }
      v4i bb = b;
      if (LDSP > 0) bb = bq[(j / LDSP + 1) % 4];
      if (LDSP < 0) bb = bw[(j / 16) & 1][(j % 16) / 2];
      if (SHAPE == 0) c[j % CHAINS] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j % 4], bb, c[j % CHAINS], 0, 0, 0);
      else d[j % CHAINS] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j % 4], bb, d[j % CHAINS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // (the MFMA stays in front of its fillers: hipcc otherwise pairs the MFMAs up)
#pragma unroll
      for (int q = 0; q < FILL; ++q) vmax_i(f[(j * FILL + q) % 16], src[(j + q) % 8]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int q = 0; q < CHAINS; ++q) s += c[q][0] + c[q][7] + d[q][1];
  for (int g = 0; g < 16; ++g) s += f[g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int CHAINS, int FILL, int LDSP> void run(int threads) {
  int* out; unsigned long long* st;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, CHAINS, FILL, LDSP>), dim3(256), dim3(threads), 0, 0, out, st, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  const double cyc = (double)h[0], real = (double)h[1];
  const double per = cyc / (iters * 32.0);
  const double ops = (SHAPE == 0 ? 32.0 * 32 * 32 : 16.0 * 16 * 64) * 2;
  const double waves = 256.0 * threads / 64;
  // SIMD-level: shader cycles of the whole launch per MFMA issued on one SIMD (both waves' MFMAs counted)
  const double clk = cyc / real * 0.1e9;
  const double per_simd = ms * 1e-3 * clk / (iters * 32.0 * (threads / 256));
  printf("%s chains=%d fill=%d lds/%d waves/SIMD=%d : wave0 %6.1f cyc/MFMA, SIMD %6.1f cyc/MFMA (pipe %d) = %4.1f%% busy  %7.1f TOP/s  clock %.2f GHz\n",
         SHAPE == 0 ? "32x32x32" : "16x16x64", CHAINS, FILL, LDSP, threads / 256, per, per_simd, SHAPE == 0 ? 32 : 16,
         100.0 * (SHAPE == 0 ? 32 : 16) / per_simd, waves * iters * 32.0 * ops / (ms * 1e-3) / 1e12, cyc / real * 0.1);
  hipFree(out); hipFree(st);
}

int main() {
  for (int th : {256, 512}) {
    run<0, 2, 0, 2>(th); run<0, 2, 1, 2>(th); run<0, 2, 2, 2>(th); run<0, 2, 3, 2>(th); run<0, 2, 4, 2>(th); run<0, 2, 5, 2>(th); run<0, 2, 6, 2>(th);
    run<0, 4, 0, 4>(th); run<0, 4, 2, 4>(th); run<0, 4, 3, 4>(th); run<0, 4, 4, 4>(th); run<0, 4, 5, 4>(th);
    run<0, 2, 0, -1>(th); run<0, 2, 3, -1>(th); run<0, 2, 4, -1>(th); run<0, 2, 5, -1>(th); run<0, 2, 6, -1>(th);
    run<1, 8, 0, 4>(th); run<1, 8, 1, 4>(th); run<1, 8, 2, 4>(th); run<1, 8, 2, -1>(th);
  }
  return 0;
}
