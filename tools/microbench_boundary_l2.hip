// Do kernel boundaries on ONE stream disturb the L2 contents another stream's kernel is living on?
// Stream A runs a kernel that re-reads a small (L2-resident) buffer many times; stream B meanwhile runs chains of
// empty kernels (each boundary = the runtime's release / acquire fences).  Compare A's duration with B idle / busy.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_boundary_l2.hip -o build/mbl2 && ./build/mbl2
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(256) void reread(const float4* __restrict__ buf, int n4, int reps, float* out) {
  float acc = 0.f;
  for (int r = 0; r < reps; ++r)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      const float4 v = buf[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 12345.f) out[0] = acc;
}
int main() {
  const int bytes = 8 << 20;                       // 8 MiB: 1 MiB per XCD's L2 share of the grid
  float4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  float* out; hipMalloc(&out, 4);
  int* d; hipMalloc(&d, 4);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int busy = 0; busy < 3; ++busy) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      if (busy) for (int i = 0; i < (busy == 1 ? 400 : 4000); ++i) hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, sb, d);
      hipEventRecord(e0, sa);
      hipLaunchKernelGGL(reread, dim3(1024), dim3(256), 0, sa, buf, bytes / 16, 200, out);
      hipEventRecord(e1, sa);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const bool still = hipStreamQuery(sb) == hipErrorNotReady;
      printf("other stream %-22s: re-reading 8 MiB x 200 took %.3f ms (%.1f TB/s)%s\n",
             busy == 0 ? "idle" : (busy == 1 ? "400 empty kernels" : "4000 empty kernels"), ms, 200.0 * bytes / ms / 1e9,
             busy ? (still ? "  [other stream still busy at the end]" : "  [other stream finished earlier]") : "");
      hipDeviceSynchronize();
    }
  }
  return 0;
}
