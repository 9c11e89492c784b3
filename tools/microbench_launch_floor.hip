#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void touch(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main() {
  int* d; hipMalloc(&d, 1 << 24);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto bench = [&](const char* name, auto fn, int n) {
    for (int w = 0; w < 20; ++w) fn();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st); for (int i = 0; i < n; ++i) fn(); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-44s %.2f us per launch\n", name, ms * 1e3 / n);
  };
  bench("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, st, d); }, 2000);
  bench("empty <<<256,256>>>", [&] { hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, st, d); }, 2000);
  bench("empty <<<247,512>>> 132KB dyn LDS", [&] { hipLaunchKernelGGL(empty, dim3(247), dim3(512), 135168, st, d); }, 2000);
  hipFuncSetAttribute((const void*)empty, hipFuncAttributeMaxDynamicSharedMemorySize, 135168);
  bench("  same, after SetAttribute", [&] { hipLaunchKernelGGL(empty, dim3(247), dim3(512), 135168, st, d); }, 2000);
  bench("  SetAttribute + launch each time", [&] { hipFuncSetAttribute((const void*)empty, hipFuncAttributeMaxDynamicSharedMemorySize, 135168); hipLaunchKernelGGL(empty, dim3(247), dim3(512), 135168, st, d); }, 2000);
  bench("touch 1M ints <<<4096,256>>>", [&] { hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, d, 1 << 20); }, 2000);
  bench("hipMemsetAsync 40 KB", [&] { hipMemsetAsync(d, 0, 40960, st); }, 2000);
  // graph of 10 empty kernels
  hipGraph_t g; hipGraphExec_t ge; hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, st, d);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  bench("graph of 10 empty <<<256,256>>> (per graph)", [&] { hipGraphLaunch(ge, st); }, 500);
  return 0;
}
