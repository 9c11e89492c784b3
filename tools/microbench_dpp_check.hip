#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
template <int MASK> __device__ float partner(float v) {
  if constexpr (MASK == 1) return dpp_mov<0xB1, 0xf>(v, v);
  else if constexpr (MASK == 2) return dpp_mov<0x4E, 0xf>(v, v);
  else if constexpr (MASK == 4) { float t = dpp_mov<0x104, 0x5>(v, v); t = dpp_mov<0x114, 0xA>(t, v); return t; }
  else if constexpr (MASK == 8) return dpp_mov<0x128, 0xf>(v, v);
  else if constexpr (MASK == 16) { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a * 1000.f + b; }
  else { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a * 1000.f + b; }
}
__global__ void k(float* out) {
  float v = (float)threadIdx.x;
  out[0 * 64 + threadIdx.x] = partner<1>(v);
  out[1 * 64 + threadIdx.x] = partner<2>(v);
  out[2 * 64 + threadIdx.x] = partner<4>(v);
  out[3 * 64 + threadIdx.x] = partner<8>(v);
  out[4 * 64 + threadIdx.x] = partner<16>(v);
  out[5 * 64 + threadIdx.x] = partner<32>(v);
}
int main() {
  float* d; hipMalloc(&d, 6 * 64 * 4); k<<<1, 64>>>(d); float h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int masks[6] = {1, 2, 4, 8, 16, 32};
  for (int m = 0; m < 6; ++m) { printf("mask %2d:", masks[m]); for (int l = 0; l < 64; ++l) printf(" %g", h[m * 64 + l]); printf("\n"); }
  return 0;
}
