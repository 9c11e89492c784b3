// Upper bound for a 128-row-per-wave form of the int8 max pass at ONE wave per SIMD: per 32-column
// unit 32 x v_mfma_i32_32x32x32_i8 on four accumulators (each B fragment feeds four MFMAs), the unit's 8 B fragments read
// from LDS in ONE burst at its head (interleaved reads and vector instructions cost ~2 cycles per filler and gap:
// tools/microbench_i8_gap.hip), FILL integer maxima per gap that read the OTHER accumulator set, one LDS-DMA piece every
// DMAP MFMAs (0 = none) and an s_barrier every 64 MFMAs (BAR).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_i8_w128.hip -o build/mb/w128 && build/mb/w128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void vmax_i(int& d, int s) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(d) : "v"(s)); }

template <int FILL, int DMAP, int BAR, int BURST, int NB>
__global__ __launch_bounds__(256, 1) void k(int* out, const char* gsrc, unsigned long long* stamps, int iters, int seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KiB ring
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v4i a[NB][8];
  for (int q = 0; q < NB; ++q)
    for (int ks = 0; ks < 8; ++ks)
      for (int i = 0; i < 4; ++i) a[q][ks][i] = seed * (threadIdx.x % 7 + i + q + ks) * 0x01010101;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)smem)[i] = i * seed;
  __syncthreads();
  v16i X[NB], Y[NB];
  for (int q = 0; q < NB; ++q) for (int g = 0; g < 16; ++g) { X[q][g] = 0; Y[q][g] = seed + g + q; }
  int f[16 * NB];
  for (int g = 0; g < 16 * NB; ++g) f[g] = seed + g;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
  v4i bw[8];            // ONE unit's fragments: k-steps 4..7 are re-read at the unit's head, 0..3 (of the next unit) at its middle
  for (int q = 0; q < 8; ++q) bw[q] = a[0][q];
  const char* gp = gsrc + (size_t)blockIdx.x * 65536 + wv * 4096 + lane * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {            // two units = one 64-column tile = 64 MFMAs
      v16i (&acc)[NB] = u ? Y : X;
      v16i (&old)[NB] = u ? X : Y;
#pragma unroll
      for (int j = 0; j < 8 * NB; ++j) {
        const int ks = j / NB, blk = j % NB;
        __builtin_amdgcn_sched_barrier(0);
        if (BURST) {
          if (j == 0) {
#pragma unroll
            for (int q = 4; q < 8; ++q)
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bw[q]) : "v"(lds0 + ((it + u) & 3) * 16384), "n"(q * 1024));
          }
          if (j == 4 * NB) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bw[q]) : "v"(lds0 + ((it + u + 1) & 3) * 16384), "n"(q * 1024));
          }
          if (j == 4 * NB - 1 || j == 8 * NB - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (blk == NB - 1) {      // one read per k-step, behind the k-step's last MFMA (the fragment 4 k-steps ahead)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bw[(ks + 4) & 7]) : "v"(lds0 + ((it + u) & 3) * 16384), "n"(((ks + 4) & 7) * 1024));
          asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        }
        if (DMAP && j % DMAP == DMAP - 1 && u == 0) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 4 + j / DMAP) & 3) * 1024),
                                           (__attribute__((address_space(3))) void*)(smem + ((it + 3) & 3) * 16384 + wv * 4096 + (j / DMAP) * 1024), 16, 0, 0);
        }
#ifdef FM_MB_ASM_MFMA      // accumulators in VGPRs (the fillers read them), A fragments in AGPRs (only MFMAs read them)
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[blk]) : "a"(a[blk][ks]), "v"(bw[ks]));
#else
        acc[blk] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[blk][ks], bw[ks], acc[blk], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < FILL; ++q) {
          const int n = (j * FILL + q) % (16 * NB);
          vmax_i(f[n], old[n / 16][n % 16]);
        }
      }
      if (BAR && u == 0) {
        if (DMAP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMAP ? 2 * (8 * NB / (DMAP ? DMAP : 1)) : 0) : "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  int s = 0;
  for (int q = 0; q < NB; ++q) s += X[q][0] + X[q][7] + Y[q][3];
  for (int g = 0; g < 16 * NB; ++g) s += f[g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int FILL, int DMAP, int BAR, int BURST, int NB> void run(const char* gsrc) {
  int* out; unsigned long long* st;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&st, 256 * 16);
  const int iters = 1000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<FILL, DMAP, BAR, BURST, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FILL, DMAP, BAR, BURST, NB>), dim3(256), dim3(256), 65536, 0, out, gsrc, st, iters, 3);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[512]; (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
  const double per = cyc / 256 / (iters * 16.0 * NB);
  printf("rows/wave=%d fill=%d dma/%d barrier=%d burst=%d : %6.1f cyc/MFMA = %4.1f%% busy   %7.1f TOP/s  clock %.2f GHz  err=%d\n", 32 * NB, FILL, DMAP, BAR, BURST, per,
         3200.0 / per, 1024.0 * iters * 16.0 * NB * 65536.0 / (ms * 1e-3) / 1e12, cyc / real * 0.1, (int)hipGetLastError());
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  char* g; (void)hipMalloc(&g, 256 * 65536); (void)hipMemset(g, 1, 256 * 65536);
  run<0, 0, 0, 1, 3>(g); run<3, 0, 0, 1, 3>(g); run<4, 0, 0, 1, 3>(g); run<5, 0, 0, 1, 3>(g); run<4, 0, 0, 0, 3>(g);
  run<4, 0, 1, 1, 3>(g); run<4, 6, 1, 1, 3>(g); run<4, 6, 1, 0, 3>(g); run<3, 6, 1, 1, 3>(g); run<5, 6, 1, 1, 3>(g);
  run<0, 0, 0, 1, 4>(g); run<3, 0, 0, 1, 4>(g); run<4, 0, 0, 1, 4>(g); run<4, 8, 1, 1, 4>(g); run<3, 8, 1, 1, 4>(g);
  run<4, 4, 1, 1, 2>(g); run<4, 4, 1, 0, 2>(g); run<5, 4, 1, 1, 2>(g);
  return 0;
}
