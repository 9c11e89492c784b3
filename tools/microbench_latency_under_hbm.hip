// Does an HBM-bound kernel on one stream stretch a latency-bound kernel on another?  X = 152 workgroups chasing
// pointers (dependent loads: every hop a memory round trip), Y = a 40 MB -> 40 MB copy on 1200 workgroups.  Each
// stream replays a hipGraph of 8 launches of its kernel (no host in the way): times of A alone, B alone, both at once.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_latency_under_hbm.hip -o build/mlat && build/mlat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void chase(const unsigned* __restrict__ next, int hops, unsigned* out) {
  unsigned p = (blockIdx.x * 256 + threadIdx.x) * 977u % (64u << 20);
  for (int i = 0; i < hops; ++i) p = next[p];
  if (p == 0xffffffffu) out[0] = p;
}
__global__ void copy4(const float4* __restrict__ s, float4* __restrict__ d, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}
__global__ void spin(int cycles, unsigned* out) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 9999) out[0] = 1;
}
int main() {
  const size_t N = 64u << 20;
  std::vector<unsigned> h(N);
  for (size_t i = 0; i < N; ++i) h[i] = (unsigned)(((i * 2654435761ull) + 12345u) % N);
  unsigned* nx; hipMalloc(&nx, N * 4); hipMemcpy(nx, h.data(), N * 4, hipMemcpyHostToDevice);
  unsigned* o; hipMalloc(&o, 4);
  const int n4 = 40 * 1000 * 1000 / 16;
  float4 *s, *d; hipMalloc(&s, (size_t)n4 * 16); hipMalloc(&d, (size_t)n4 * 16); hipMemset(s, 0, (size_t)n4 * 16);
  hipStream_t st[2];
  hipGraphExec_t g[3][2];      // kind 0 = chase, 1 = copy, 2 = spin; per stream
  for (int k = 0; k < 2; ++k) hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
  for (int kind = 0; kind < 3; ++kind)
    for (int k = 0; k < 2; ++k) {
      hipGraph_t gr;
      hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < 8; ++i) {
        if (kind == 0) hipLaunchKernelGGL(chase, dim3(152), dim3(256), 0, st[k], nx, 48, o);
        else if (kind == 1) hipLaunchKernelGGL(copy4, dim3(1200), dim3(256), 0, st[k], s, d, n4);
        else hipLaunchKernelGGL(spin, dim3(152), dim3(256), 0, st[k], 20000, o);
      }
      hipStreamEndCapture(st[k], &gr);
      hipGraphInstantiate(&g[kind][k], gr, nullptr, nullptr, 0);
    }
  const char* names[3] = {"chase (152 wg, 48 dependent hops)", "copy 40 MB (1200 wg)", "spin ~9 us (152 wg)"};
  auto run = [&](int ka, int kb) {       // kb < 0: stream 0 alone
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 10; ++i) {
        hipGraphLaunch(g[ka][0], st[0]);
        if (kb >= 0) hipGraphLaunch(g[kb][1], st[1]);
      }
      hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 80;
      if (us < best) best = us;
    }
    return best;
  };
  for (int k = 0; k < 3; ++k) printf("%-36s alone: %.2f us per launch\n", names[k], run(k, -1));
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b)
      printf("%-36s || %-36s: %.2f us per launch pair\n", names[a], names[b], run(a, b));
  return 0;
}
