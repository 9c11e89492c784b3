// Throughput floor of dependent kernel chains replayed as hipGraphs on several streams: how many (empty) kernels per
// second the command processor / dispatch path retires when S streams each run a chain of K dependent kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_multistream_floor.hip -o /tmp/msf && /tmp/msf
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void spin(int* p, int cycles) {            // ~cycles of shader clock per wave, no memory traffic
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (p && threadIdx.x == 9999) *p = 1;
}
int main(int argc, char** argv) {
  int* d; hipMalloc(&d, 1 << 20);
  const int K = 6;
  const bool direct = argc > 1 && argv[1][0] == 'd';      // 'd': plain launches instead of graph replays
  for (int mode = 2; mode < 3; ++mode) {
    for (int S : {1, 2, 3, 4, 6, 8}) {
      std::vector<hipStream_t> st(S);
      std::vector<hipGraphExec_t> ge(S);
      for (int s = 0; s < S; ++s) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipGraph_t g;
        hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < K; ++i) {
          if (mode == 0) hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, st[s], d);
          else if (mode == 1) hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, st[s], d);
          else if (mode == 2) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[s], d, 12000);     // ~5 us on a quarter of the CUs
          else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[s], d, 12000);                   // ~5 us on every CU (1 WG each)
        }
        hipStreamEndCapture(st[s], &g);
        hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
      }
      const int R = 400;
      auto go = [&](int s) {
        if (!direct) { hipGraphLaunch(ge[s], st[s]); return; }
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[s], d, 12000);
      };
      for (int w = 0; w < 20; ++w) for (int s = 0; s < S; ++s) go(s);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < R; ++i) for (int s = 0; s < S; ++s) go(s);
      hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      const char* names[] = {"empty <<<1,64>>>", "empty <<<256,256>>>", "spin 5us <<<64,256>>>", "spin 5us <<<256,256>>>"};
      printf("%-24s %d streams x chains of %d: %.2f us per kernel (whole job), %.2f us per kernel per stream\n", names[mode], S, K,
             us / (R * S * K), us / (R * K));
      for (int s = 0; s < S; ++s) { hipGraphExecDestroy(ge[s]); hipStreamDestroy(st[s]); }
    }
  }
  return 0;
}
