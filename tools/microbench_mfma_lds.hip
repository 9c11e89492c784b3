#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// VAR 0: registers only (3 MFMA per step, distinct A regs); 1: + 2 ds_read_b128 per step via asm ring (PF 2);
// 2: same with plain C++ LDS loads; 3: like 1 but B operands constant (reads issued, results unused by MFMA)
template <int VAR, int PF>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  half8 ahi[16], alo[16];
  for (int k = 0; k < 16; ++k) for (int e = 0; e < 8; ++e) { ahi[k][e] = (_Float16)(0.01f * (lane % 7 + k + e)); alo[k][e] = (_Float16)(0.001f * (lane % 5 + k)); }
  f32x16 acc = {0};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned base = lds0 + lane * 16;
  half8 cb; for (int e = 0; e < 8; ++e) cb[e] = (_Float16)(0.02f * e);
  for (int it = 0; it < iters; ++it) {
    half8 bh[PF + 1], bl[PF + 1];
    if (VAR == 1 || VAR == 3) {
      auto issue = [&](int ks) {
        const unsigned la = base + ks * 1024;
        asm volatile("ds_read_b128 %0, %1" : "=v"(bh[ks % (PF + 1)]) : "v"(la));
        asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(bl[ks % (PF + 1)]) : "v"(la));
      };
#pragma unroll
      for (int q = 0; q < PF; ++q) issue(q);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + PF < 16) issue(ks + PF);
        const int ahead = (15 - ks) < PF ? (15 - ks) : PF;
#define W(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(bh[ks % (PF + 1)]), "+v"(bl[ks % (PF + 1)]))
        switch (ahead) { case 0: W(0); break; case 1: W(2); break; case 2: W(4); break; case 3: W(6); break; case 4: W(8); break; case 5: W(10); break; case 6: W(12); break; default: W(14); break; }
        const half8 h8 = VAR == 3 ? cb : bh[ks % (PF + 1)];
        const half8 l8 = VAR == 3 ? cb : bl[ks % (PF + 1)];
        if (VAR == 3) asm volatile("" :: "v"(bh[ks % (PF + 1)]), "v"(bl[ks % (PF + 1)]));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], l8, acc, 0, 0, 0);
      }
    } else if (VAR == 2) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const half8 h8 = *(const half8*)(smem + lane * 16 + ks * 1024);
        const half8 l8 = *(const half8*)(smem + 32768 + lane * 16 + ks * 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], h8, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], l8, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], cb, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], cb, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], cb, acc, 0, 0, 0);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[7];
}
template <int VAR, int PF> void run(const char* name, int threads) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<VAR, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 600; float ms = 0;
  for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((k<VAR, PF>), dim3(256), dim3(threads), 65536, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
  double n = 256.0 * threads / 64 * iters * 48.0;
  printf("%-46s threads=%d  %.3f ms  %.0f TFLOP/s\n", name, threads, ms, n * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
  for (int th : {256, 512}) {
    run<0, 2>("registers only (3 MFMA/step)", th);
    run<1, 1>("asm ds_read ring PF=1", th);
    run<1, 2>("asm ds_read ring PF=2", th);
    run<1, 3>("asm ds_read ring PF=3", th);
    run<1, 4>("asm ds_read ring PF=4", th);
    run<1, 6>("asm ds_read ring PF=6", th);
  }
  return 0;
}
