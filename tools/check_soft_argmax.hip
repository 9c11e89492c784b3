// Standalone check of fine.hip's cross-lane code (soft_argmax2 and the 16-value partial transpose-reduce) against plain
// __shfl_xor butterflies, on one wave: the place to look first when a DPP / permlane / readlane sequence misbehaves
// (two bugs of that kind were found with it: an add that hipcc sank behind a lane-0 branch although other lanes' values are
// read by v_readlane, and plain VALU instructions that hipcc scheduled right in front of an asm DPP read of their result).
#include "../featurematching_amd/csrc/fine.hip"   // hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ifeaturematching_amd/csrc tools/check_soft_argmax.hip -o build/check_sa && build/check_sa
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace fm;
template <int W>
__global__ void k_test(const float* s0, const float* s1, float* o0, float* o1, float* r0, float* r1) {
  constexpr int WW = W * W; constexpr int NP = WW > 32 ? 64 : 32;
  const int lane = threadIdx.x;
  const int pos = tr_index<NP>(lane);
  const bool on = pos < WW && (NP == 64 || !(lane & 1));
  const float a = on ? s0[pos] : 0.f, b = on ? s1[pos] : 0.f;
  soft_argmax2<W>(a, b, pos, on, lane, 0.125f, 2.0f, 10.f, 20.f, 30.f, 40.f, o0, o1);
  // reference: plain shuffles
  for (int d = 0; d < 2; ++d) {
    const float x = on ? (d ? b : a) * 0.125f : -INFINITY;
    float m = x; for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    const float e = on ? __expf(x - m) : 0.f;
    const int wy = pos / W, wx = pos - wy * W;
    const float gx = ((float)wx / (float)(W - 1) - 0.5f) * 2.f, gy = ((float)wy / (float)(W - 1) - 0.5f) * 2.f;
    float v[5] = {e, gx * e, gy * e, gx * gx * e, gy * gy * e};
    for (int q = 0; q < 5; ++q) for (int k = 32; k >= 1; k >>= 1) v[q] += __shfl_xor(v[q], k);
    if (lane == 0) { float* r = d ? r1 : r0; for (int q = 0; q < 5; ++q) r[q] = v[q]; }
  }
}
template <int W>
__global__ void k_dbg(const float* s0, float* tout, float* rout) {
  constexpr int WW = W * W; constexpr int NP = WW > 32 ? 64 : 32;
  const int lane = threadIdx.x;
  float q[16];
  for (int k = 0; k < 16; ++k) q[k] = k < 10 ? (float)(k + 1) * (1.0f + (float)((lane * 7 + k) % 5)) : 0.f;
  float ref[16];
  for (int k = 0; k < 16; ++k) { float v = q[k]; for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m); ref[k] = v; }
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(q[k]), "+v"(q[k + 8]));
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] += q[k + 8];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(q[k]), "+v"(q[k + 4]));
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] += q[k + 4];
  const float q3b = q[3];
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int k = 0; k < 2; ++k)
    asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(q[k]));
#pragma unroll
  for (int k = 0; k < 2; ++k)
    asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(q[k]) : "v"(q[k + 2]));
  tout[64 + lane] = q3b; tout[128 + lane] = q[1]; tout[192 + lane] = q[0];
  asm volatile("s_nop 1" ::: "memory");
  asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(q[0]));
  asm volatile("s_nop 1" ::: "memory");
  asm volatile("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(q[0]) : "v"(q[1]));
  float t = q[0];
  asm volatile("s_nop 1" : "+v"(t));
  tout[256 + lane] = t;
  t += dpp_mov<0x4E, 0xf>(t, t);
  t += dpp_mov<0xB1, 0xf>(t, t);
  tout[lane] = t;
  if (lane == 0) for (int k = 0; k < 16; ++k) rout[k] = ref[k];
}
int main() {
  { float hh[64]; for (int i = 0; i < 64; ++i) hh[i] = (float)(i % 11) + 0.25f * (i % 3);
    float *d0, *dt, *dr; hipMalloc(&d0, 256); hipMalloc(&dt, 2048); hipMalloc(&dr, 64);
    hipMemcpy(d0, hh, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dbg<5>, dim3(1), dim3(64), 0, 0, d0, dt, dr);
    float ht[320], hr[16]; hipMemcpy(ht, dt, 1280, hipMemcpyDeviceToHost); hipMemcpy(hr, dr, 64, hipMemcpyDeviceToHost);
    printf("ref:"); for (int k = 0; k < 16; ++k) printf(" %.2f", hr[k]); printf("\n");
    printf("lane: q3 before C | q1 after C | q0 after C | q0 after D\n");
    for (int l = 0; l < 16; ++l) printf("%2d: %7.2f %7.2f %7.2f %7.2f\n", l, ht[64 + l], ht[128 + l], ht[192 + l], ht[256 + l]);
    for (int l = 0; l < 64; l += 4) printf("lane %2d (idx %2d): %.2f\n", l, 8 * ((l >> 5) & 1) + 4 * ((l >> 4) & 1) + 2 * ((l >> 3) & 1) + ((l >> 2) & 1), ht[l]); }

  float h0[64], h1[64]; for (int i = 0; i < 64; ++i) { h0[i] = (rand() % 1000) / 50.f; h1[i] = (rand() % 1000) / 50.f; }
  float *s0, *s1, *o0, *o1, *r0, *r1;
  hipMalloc(&s0, 256); hipMalloc(&s1, 256); hipMalloc(&o0, 64); hipMalloc(&o1, 64); hipMalloc(&r0, 64); hipMalloc(&r1, 64);
  hipMemcpy(s0, h0, 256, hipMemcpyHostToDevice); hipMemcpy(s1, h1, 256, hipMemcpyHostToDevice);
  for (int w : {5, 7}) {
    if (w == 5) hipLaunchKernelGGL(k_test<5>, dim3(1), dim3(64), 0, 0, s0, s1, o0, o1, r0, r1);
    else hipLaunchKernelGGL(k_test<7>, dim3(1), dim3(64), 0, 0, s0, s1, o0, o1, r0, r1);
    float a[3], b[3], ra[5], rb[5];
    hipMemcpy(a, o0, 12, hipMemcpyDeviceToHost); hipMemcpy(b, o1, 12, hipMemcpyDeviceToHost);
    hipMemcpy(ra, r0, 20, hipMemcpyDeviceToHost); hipMemcpy(rb, r1, 20, hipMemcpyDeviceToHost);
    for (int d = 0; d < 2; ++d) {
      const float* r = d ? rb : ra; const float* o = d ? b : a;
      const float cx = r[1] / r[0], cy = r[2] / r[0], vx = r[3] / r[0] - cx * cx, vy = r[4] / r[0] - cy * cy;
      const float kx = d ? 30.f : 10.f, ky = d ? 40.f : 20.f;
      printf("W=%d dir %d: got %.5f %.5f %.5f  want %.5f %.5f %.5f\n", w, d, o[0], o[1], o[2],
             kx + (cx * (w / 2) * 2.0f + (w / 2)), ky + (cy * (w / 2) * 2.0f + (w / 2)), sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f)));
    }
  }
  return 0;
}
