#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: one dependent chain of 32x32x16 per wave; 1: two independent chains; 2: 16x16x32 chain x4 independent
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters, float seed) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (threadIdx.x % 7 + i) * 0.01f); b[i] = (_Float16)(seed * (threadIdx.x % 5 + i) * 0.02f); }
  f32x16 c0 = {0}, c1 = {0};
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d1, 0, 0, 0); d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, d3, 0, 0, 0); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = c0[0] + c1[3] + d0[0] + d1[1] + d2[2] + d3[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int MODE> void run(const char* name, int threads, int iters, int mfma_per_iter, double flop_per_mfma) {
  float* out; unsigned long long* st; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, st, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double waves = 256.0 * threads / 64; double n = waves * iters * (double)mfma_per_iter;
  double cyc = (double)h[0], real = (double)h[1];
  printf("%-34s threads=%d  %.3f ms  %.1f TFLOP/s  cycles/MFMA/wave=%.1f  clock=%.2f GHz\n", name, threads, ms, n * flop_per_mfma / (ms * 1e-3) / 1e12,
         cyc / (iters * (double)mfma_per_iter), cyc / real * 0.1);
}
int main() {
  for (int th : {256, 512}) {
    run<0>("32x32x16 f16, 1 chain", th, 4000, 16, 32.0 * 32 * 16 * 2);
    run<1>("32x32x16 f16, 2 chains", th, 4000, 16, 32.0 * 32 * 16 * 2);
    run<2>("16x16x32 f16, 4 chains", th, 4000, 16, 16.0 * 16 * 32 * 2);
  }
  return 0;
}
