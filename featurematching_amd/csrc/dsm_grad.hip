// Training surface of the coarse stage: the dual softmax at chosen entries and its backward,
// without any [N, L, S] temporary.
//
// The reference's coarse loss (losses/loss.py:27-67, `sparse_spvs`: the default) reads data['conf_matrix'] only at the
// supervised entries; its gradient w.r.t. the descriptors goes through conf = A * B, A = softmax(sim, dim 1) (over i),
// B = softmax(sim, dim 2) (over j), sim = f0 . f1^T / (C T) (network/utils/coarse_matching_new.py:64-68).  With
// g_e = dL/dconf_e at the supervised entries e = (b, i, j) and c_e = conf_e:
//     dL/dsim_kl = 2 g c [kl supervised] - A_kl u_l - B_kl v_k,      u_l = sum_e[j_e = l] g_e c_e,   v_k = sum_e[i_e = k] g_e c_e
//     dL/df0 = dL/dsim . f1 / (C T),     dL/df1 = dL/dsim^T . f0 / (C T)
// A and B follow from the softmax statistics the coarse stage leaves in its workspace (stabilisers and denominators:
// A_kl = exp2(k2 x_kl + nm_c[l]) / sum_c[l], B_kl = exp2(k2 x_kl + nm_r[k]) / sum_r[k], x = raw dot product, k2 = log2(e) /
// (C T); kept apart - folded into one offset nm - log2(sum) the float32 rounding of ~230 - 230 would cost 1e-5 of a conf
// near 1), so the backward is two
// launches of ONE kernel with the roles of the images swapped: a workgroup owns 32 rows of the "owner" image, sweeps the
// other image in tiles of 32 descriptors, recomputes the 32 x 32 similarities of the tile in float32 (exact products,
// fused multiply-adds in channel order), turns them into D = -(A u + B v) and accumulates D . other into its rows'
// gradient - the dense matrix never exists.  The supervised entries' own term 2 g c is K rows: a third, tiny kernel.
// float32 vector arithmetic on purpose: a training-only path whose bar is the agreement with float64 autograd
// (tests), not the matrix cores.
#include "fm_device.h"

namespace fm {

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov_g(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float row_sum16_g(float v) {      // sum over the 16 lanes of a DPP row, fixed order
  v = v + dpp_mov_g<0xB1, 0xf>(v, v);
  v = v + dpp_mov_g<0x4E, 0xf>(v, v);
  v = v + dpp_mov_g<0x141, 0xf>(v, v);
  v = v + dpp_mov_g<0x140, 0xf>(v, v);
  return v;
}

// conf at K entries: 16 lanes per entry (16 channels per lane, four 16-byte loads per row), the exact float32 dot
// product in a fixed order - the arithmetic of k_screen's exact phase.
__global__ __launch_bounds__(256) void k_conf_at(const float* __restrict__ f0, const float* __restrict__ f1, int L, int S, int c_in,
                                                 float k2, const float* __restrict__ nm_r, const float* __restrict__ sum_r,
                                                 int pitch_r, const float* __restrict__ nm_c,
                                                 const float* __restrict__ sum_c, int pitch_c,
                                                 const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                                 const int64_t* __restrict__ j_ids, int K, float* __restrict__ conf,
                                                 float* __restrict__ xdot) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
  const int ec = min(e, K - 1);
  const long b = b_ids[ec], i = i_ids[ec], j = j_ids[ec];
  const float4* ra = reinterpret_cast<const float4*>(f0 + (b * L + i) * c_in);
  const float4* rb = reinterpret_cast<const float4*>(f1 + (b * S + j) * c_in);
  const int vpr = c_in >> 2;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v4 = l16 + 16 * q;
    const bool in = v4 < vpr;
    const float4 a = ra[in ? v4 : 0], bb = rb[in ? v4 : 0];
    const float4 az = in ? a : make_float4(0.f, 0.f, 0.f, 0.f);
    s = __builtin_fmaf(az.x, bb.x, s);
    s = __builtin_fmaf(az.y, bb.y, s);
    s = __builtin_fmaf(az.z, bb.z, s);
    s = __builtin_fmaf(az.w, bb.w, s);
  }
  const float x = row_sum16_g(s);
  if (l16 == 0 && e < K) {
    conf[e] = (__builtin_amdgcn_exp2f(__builtin_fmaf(x, k2, nm_r[b * pitch_r + i])) / sum_r[b * pitch_r + i]) *
              (__builtin_amdgcn_exp2f(__builtin_fmaf(x, k2, nm_c[b * pitch_c + j])) / sum_c[b * pitch_c + j]);
    if (xdot) xdot[e] = x;
  }
}

// u_l and v_k of the supervised entries (float atomics: entries that share a row or a column are rare, and the order
// in which two of them are added is the only thing that is not fixed)
__global__ __launch_bounds__(256) void k_dsm_uv(const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                                const int64_t* __restrict__ j_ids, const float* __restrict__ gc, int K,
                                                int L, int S, float* __restrict__ v, float* __restrict__ u) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= K) return;
  atomicAdd(&v[b_ids[e] * L + i_ids[e]], gc[e]);
  atomicAdd(&u[b_ids[e] * S + j_ids[e]], gc[e]);
}

// dX[k, :] = sum_l D_kl Y[l, :],  D_kl = -(exp2(k2 x_kl + nm_y[l]) w_y[l] / sum_y[l] + exp2(k2 x_kl + nm_x[k]) w_x[k] / sum_x[k]),
// x = X_k . Y_l
// grid (ceil(R / 32), N, Z): workgroup = 32 owner rows x the z-th share of the other image's 32-descriptor tiles.
// C = padded channel count (64 / 128 / 256), c_in <= C the rows' real length.  Partial gradients (one per z) go to
// part[z][b][row][c_in]; k_dsm_combine adds them up and scales.
// MODE (round 5: a DENSE dL/dconf = G [N, L, S], what the reference's loss over all negatives hands back - losses/loss.py:44-50,
// 62-65 - without any [N, L, S] temporary):
//   kDsmSparse : as above (G lives in w_x / w_y and the entries' own kernel).
//   kDsmStats  : v_k = sum_l G_kl conf_kl and u_l = sum_k G_kl conf_kl with conf recomputed tile by tile from exact float32
//                dot products (conf = A B); no gradient phase.  Launched on side 0 only; v and u by float atomics: a row's
//                v takes one add per z slice (<= 4), a COLUMN's u one add per 32-row tile of the owner image (150 at
//                640x480) in arrival order - the dense backward's gradients are therefore reproducible to float32
//                rounding of those sums (~1e-7 relative), not bit for bit; every other reduction of this file is
//                order-fixed (k_fix_sums, k_dsm_combine).
//   kDsmDense  : D_kl = 2 G_kl conf_kl - A_kl u_l - B_kl v_k, the whole of dL/dsim, accumulated into the rows' gradient.
// G is read through a transposing LDS tile when the owner image is image 1 (g_t: element (owner row, other row) lives at
// G[other][owner]): both sides read 128-byte row segments of G.  G is read three times in all (92 MB per 640x480 pair
// each), nothing of its size is written.
enum { kDsmSparse = 0, kDsmStats = 1, kDsmDense = 2 };
template <int C, int MODE>
__global__ __launch_bounds__(256) void k_dsm_bwd(const float* __restrict__ X, const float* __restrict__ Y, int R, int T, int c_in,
                                                 const float* __restrict__ ofs_x, const float* __restrict__ sum_x, int pitch_x,
                                                 const float* __restrict__ ofs_y, const float* __restrict__ sum_y,
                                                 int pitch_y, const float* __restrict__ w_x, const float* __restrict__ w_y,
                                                 float k2, float* __restrict__ part, const float* __restrict__ G, int g_t,
                                                 float* __restrict__ v_out, float* __restrict__ u_out) {
  constexpr int P = C + 4;                   // row pitch (floats): 16-byte reads of 16 consecutive rows hit all banks
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Xs = sm;                            // [32][P]
  float* Ys = sm + 32 * P;                   // [32][P]
  float* Dt = sm + 64 * P;                   // [32 (l)][36]: D transposed
  const int tid = threadIdx.x, b = blockIdx.y, k0 = blockIdx.x * 32;
  const int ntiles = (T + 31) / 32, Z = gridDim.z, z = blockIdx.z;
  const int t_lo = (int)((long)ntiles * z / Z), t_hi = (int)((long)ntiles * (z + 1) / Z);
  const float* Xb = X + (long)b * R * c_in;
  const float* Yb = Y + (long)b * T * c_in;
  const int vpr = c_in >> 2;
  auto load_tile = [&](float* dst, const float* src, int row0, int rows) {
#pragma unroll
    for (int p = 0; p < 32 * (C / 4) / 256; ++p) {
      const int idx = p * 256 + tid, row = idx / (C / 4), v4 = idx % (C / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < rows && v4 < vpr) v = reinterpret_cast<const float4*>(src + (long)(row0 + row) * c_in)[v4];
      *reinterpret_cast<float4*>(&dst[row * P + 4 * v4]) = v;
    }
  };
  load_tile(Xs, Xb, k0, R);
  const int tx = tid & 31, ty = tid >> 5;                 // similarity phase: column tx, rows ty + 8 q
  float ox[4], wx[4], isx[4], vacc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = k0 + ty + 8 * q;
    ox[q] = k < R ? ofs_x[(long)b * pitch_x + k] : 0.f;
    isx[q] = k < R ? 1.0f / sum_x[(long)b * pitch_x + k] : 0.f;
    if (MODE == kDsmSparse) wx[q] = k < R ? w_x[(long)b * R + k] / sum_x[(long)b * pitch_x + k] : 0.f;
    else wx[q] = (MODE == kDsmDense && k < R) ? w_x[(long)b * R + k] : 0.f;
    vacc[q] = 0.f;
  }
  float* Gs = Dt + 32 * 36;                  // [32][33]: the tile of G (MODE != kDsmSparse)
  const float* Gb = G ? G + (long)b * (g_t ? (long)T * R : (long)R * T) : nullptr;
  const int c4 = tid & 63, rg = tid >> 6;                 // gradient phase: channels 4 c4 .. + 3, rows 8 rg .. + 7
  float acc[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
  for (int t = t_lo; t < t_hi; ++t) {
    const int l0 = t * 32;
    __syncthreads();                                      // the previous tile's readers are done with Ys and Dt
    load_tile(Ys, Yb, l0, T);
    const int l = l0 + tx;
    const float oy = l < T ? ofs_y[(long)b * pitch_y + l] : 0.f;
    const float isy = l < T ? 1.0f / sum_y[(long)b * pitch_y + l] : 0.f;
    float wy;
    if (MODE == kDsmSparse) wy = l < T ? w_y[(long)b * T + l] / sum_y[(long)b * pitch_y + l] : 0.f;
    else wy = (MODE == kDsmDense && l < T) ? w_y[(long)b * T + l] : 0.f;
    float gq[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != kDsmSparse) {
      // 128-byte row segments of G either way: g_t = 0 -> G[owner k0 + ty + 8q][other l0 + tx] straight into registers;
      // g_t = 1 -> G[other l0 + ty + 8q][owner k0 + tx] into the LDS tile, read back transposed behind the barrier
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!g_t) {
          const int k = k0 + ty + 8 * q;
          gq[q] = (k < R && l < T) ? Gb[(long)k * T + l] : 0.f;
        } else {
          const int lo = l0 + ty + 8 * q, ko = k0 + tx;
          Gs[(ty + 8 * q) * 33 + tx] = (lo < T && ko < R) ? Gb[(long)lo * R + ko] : 0.f;
        }
      }
    }
    __syncthreads();
    if (MODE != kDsmSparse && g_t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gq[q] = Gs[tx * 33 + ty + 8 * q];
    }
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < C; c += 4) {
      const float4 y = *reinterpret_cast<const float4*>(&Ys[tx * P + c]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(&Xs[(ty + 8 * q) * P + c]);
        sv[q] = __builtin_fmaf(x.x, y.x, sv[q]);
        sv[q] = __builtin_fmaf(x.y, y.y, sv[q]);
        sv[q] = __builtin_fmaf(x.z, y.z, sv[q]);
        sv[q] = __builtin_fmaf(x.w, y.w, sv[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = l < T && k0 + ty + 8 * q < R;
      if (MODE == kDsmSparse) {
        const float a = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[q], k2, oy)) * wy;
        const float bt = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[q], k2, ox[q])) * wx[q];
        Dt[tx * 36 + ty + 8 * q] = ok ? -(a + bt) : 0.f;
      } else {
        const float ar = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[q], k2, oy)) * isy;       // softmax over the owner's rows
        const float br = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[q], k2, ox[q])) * isx[q]; // ... over the other image's
        const float gc = ok ? gq[q] * (ar * br) : 0.f;
        if (MODE == kDsmStats) { vacc[q] += gc; Dt[tx * 36 + ty + 8 * q] = gc; }
        else Dt[tx * 36 + ty + 8 * q] = ok ? 2.0f * gc - ar * wy - br * wx[q] : 0.f;
      }
    }
    __syncthreads();
    if (MODE == kDsmStats) {
      // column sums of this tile's G conf: 32 threads add the 32 owner rows of their column in a fixed order
      if (tid < 32 && l0 + tid < T) {
        float cs = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) cs += Dt[tid * 36 + rr];
        atomicAdd(&u_out[(long)b * T + l0 + tid], cs);
      }
      continue;
    }
    if (4 * c4 < C) {
#pragma unroll 4
      for (int ll = 0; ll < 32; ++ll) {
        const float4 y = *reinterpret_cast<const float4*>(&Ys[ll * P + 4 * c4]);
        const float4 d0 = *reinterpret_cast<const float4*>(&Dt[ll * 36 + 8 * rg]);
        const float4 d1 = *reinterpret_cast<const float4*>(&Dt[ll * 36 + 8 * rg + 4]);
        const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          acc[r][0] = __builtin_fmaf(d[r], y.x, acc[r][0]);
          acc[r][1] = __builtin_fmaf(d[r], y.y, acc[r][1]);
          acc[r][2] = __builtin_fmaf(d[r], y.z, acc[r][2]);
          acc[r][3] = __builtin_fmaf(d[r], y.w, acc[r][3]);
        }
      }
    }
  }
  if (MODE == kDsmStats) {
    // row sums: the 32 columns a row's partial sums sit in are the 32 lanes of a half wave
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = vacc[q];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) t += __shfl_xor(t, m);
      const int k = k0 + ty + 8 * q;
      if (tx == 0 && k < R) atomicAdd(&v_out[(long)b * R + k], t);
    }
    return;
  }
  if (c4 < vpr) {
    float* out = part + (((long)z * gridDim.y + b) * R) * c_in;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = k0 + 8 * rg + r;
      if (k < R) reinterpret_cast<float4*>(out + (long)k * c_in)[c4] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
}

// out = scale * sum_z part[z]   (fixed order)
__global__ __launch_bounds__(256) void k_dsm_combine(const float4* __restrict__ part, long n4, int Z, float scale,
                                                     float4* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = part[i];
  for (int z = 1; z < Z; ++z) {
    const float4 v = part[(long)z * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[i] = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
}

// the supervised entries' own term: d0[b, i, :] += 2 g c / (C T) f1[b, j, :],  d1[b, j, :] += 2 g c / (C T) f0[b, i, :]
// (one wave per entry; float atomics - see k_dsm_uv)
__global__ __launch_bounds__(256) void k_dsm_entries(const float* __restrict__ f0, const float* __restrict__ f1, int L, int S,
                                                     int c_in, const int64_t* __restrict__ b_ids,
                                                     const int64_t* __restrict__ i_ids, const int64_t* __restrict__ j_ids,
                                                     const float* __restrict__ gc, int K, float scale2,
                                                     float* __restrict__ d0, float* __restrict__ d1) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= K) return;
  const long b = b_ids[e], i = i_ids[e], j = j_ids[e];
  const float w = gc[e] * scale2;
  for (int c = lane; c < c_in; c += 64) {
    atomicAdd(&d0[(b * L + i) * c_in + c], w * f1[(b * S + j) * c_in + c]);
    atomicAdd(&d1[(b * S + j) * c_in + c], w * f0[(b * L + i) * c_in + c]);
  }
}

// Dense data['conf_matrix'] (k_dense<C, CONF>: hi/lo-split float16 products, 22 significant bits - 2e-4 in a conf near 1 at
// |sim| ~ 200): every entry that matters takes the exact float32 route.  For the samples the screening kernel served,
// the rows' lists hold every entry with a row term above 2^-32 (all entries with conf > 2.4e-10 are among them) together
// with its exact dot product; their conf is rewritten from that number and the same log-softmax offsets.
// One thread per (row, slot).
__global__ __launch_bounds__(256) void k_conf_patch(const int* __restrict__ rcount, const int* __restrict__ rlist_j,
                                                    const float* __restrict__ rlist_x, const int* __restrict__ rcount_d,
                                                    const int* __restrict__ rlist_j_d, const float* __restrict__ rlist_x_d,
                                                    const float* __restrict__ nm_r,
                                                    const float* __restrict__ sum_r, const float* __restrict__ nm_c,
                                                    const float* __restrict__ sum_c, const int* __restrict__ dense_cnt, int N,
                                                    int L, int S, int Lp, int Sp, int slots, float k2, float* __restrict__ conf) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long grow = gid / slots;
  const int slot = (int)(gid - grow * slots);
  if (grow >= (long)N * Lp) return;
  const int b = (int)(grow / Lp), i = (int)(grow - (long)b * Lp);
  if (i >= L) return;
  const bool dense = dense_cnt[b] > 0;         // the dense kernel's lists, made exact by k_exact_lists
  if (slot >= min((dense ? rcount_d : rcount)[grow], slots)) return;
  const float x = (dense ? rlist_x_d : rlist_x)[grow * slots + slot];
  if (!(x > -INFINITY)) return;                    // a reserved but empty place
  const int j = (dense ? rlist_j_d : rlist_j)[grow * slots + slot];
  const long gcol = (long)b * Sp + j;
  conf[((long)b * L + i) * S + j] = (__builtin_amdgcn_exp2f(__builtin_fmaf(x, k2, nm_r[grow])) / sum_r[grow]) *
                                    (__builtin_amdgcn_exp2f(__builtin_fmaf(x, k2, nm_c[gcol])) / sum_c[gcol]);
}

// The samples the DENSE kernel served (flat similarity somewhere in the sample), when every entry is read (conf_matrix,
// softmax statistics): the hi/lo-split float16 products carry 22 bits, 2^-22 |sim| ~ 4e-5 at similarities of ~170 - and
// a peaked row of such a sample (conf ~ 1: the entry IS its row's and its column's denominator) is only right when the
// entry and the two denominators hold the SAME x.  So the lists of such a sample - with a conf_matrix request they are
// formed with min(thr, 0.1): every entry whose two softmax factors both exceed 0.1, i.e. every entry with conf > 0.1 -
// are resolved exactly, entries and denominators together:
//   k_exact_lists : 16 lanes per row list and per column list, its entries in turn: the exact float32 dot product of the caller's
//                   descriptors (the arithmetic of k_conf_at) replaces the list's x; the term's change
//                   exp2(k x + nm) - exp2(k x22 + nm) goes to the screening kernel's list region of the sample (idle)
//   k_fix_sums    : one thread per row / column: adds the changes in index order (same bits every run) to the
//                   denominator, refreshes the log-softmax offset
// The assignment then reads exact x and matching denominators (mconf = the conf_matrix entry), k_conf_patch rewrites the
// listed entries of the dense matrix.  What stays 22-bit has conf <= 0.1: 2 * 2^-22 |sim| conf <= 8e-6 at |sim| ~ 170.
__global__ __launch_bounds__(256) void k_exact_lists(const void* __restrict__ f0, const void* __restrict__ f1, int in_dtype, int c_in,
                                                     const int* __restrict__ rcount, const int* __restrict__ rkey, float* __restrict__ rx,
                                                     float* __restrict__ rdelta, const int* __restrict__ ccount,
                                                     const int* __restrict__ ckey, float* __restrict__ cx, float* __restrict__ cdelta,
                                                     const float* __restrict__ nm_r, const float* __restrict__ nm_c,
                                                     const int* __restrict__ dense_cnt, int N, int L, int S, int Lp, int Sp,
                                                     int slots, float k2) {
  const int side = blockIdx.z;
  const int len = side ? Sp : Lp;
  const long gl = (long)blockIdx.x * 16 + (threadIdx.x >> 4);           // one 16-lane group per list, its slots in turn
  const int l16 = threadIdx.x & 15;
  if (gl >= (long)N * len) return;
  const int b = (int)(gl / len), own = (int)(gl - (long)b * len);
  if (own >= (side ? S : L) || dense_cnt[b] == 0) return;              // (uniform over the group)
  const int n = min((side ? ccount : rcount)[gl], slots);
  const int vpr = c_in >> 2;
  for (int slot = 0; slot < n; ++slot) {
    const long at = gl * slots + slot;
    const int other = (side ? ckey : rkey)[at];
    const int i = side ? other : own, j = side ? own : other;
    const long ro = ((long)b * L + i) * c_in, co = ((long)b * S + j) * c_in;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int v4 = l16 + 16 * q;
      const bool in = v4 < vpr;
      const int vc = in ? v4 : 0;
      float4 a, bb;
      if (in_dtype == FM_F32) {
        a = reinterpret_cast<const float4*>((const float*)f0 + ro)[vc];
        bb = reinterpret_cast<const float4*>((const float*)f1 + co)[vc];
      } else {
        a = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)f0 + ro)[vc], in_dtype);
        bb = half4_to_float4(reinterpret_cast<const uint2*>((const unsigned short*)f1 + co)[vc], in_dtype);
      }
      if (!in) a = make_float4(0.f, 0.f, 0.f, 0.f);
      s = __builtin_fmaf(a.x, bb.x, s);
      s = __builtin_fmaf(a.y, bb.y, s);
      s = __builtin_fmaf(a.z, bb.z, s);
      s = __builtin_fmaf(a.w, bb.w, s);
    }
    const float x = row_sum16_g(s);
    if (l16 == 0) {
      float* xs = side ? cx : rx;
      const float nm = (side ? nm_c : nm_r)[gl];
      (side ? cdelta : rdelta)[at] = __builtin_amdgcn_exp2f(__builtin_fmaf(x, k2, nm)) - __builtin_amdgcn_exp2f(__builtin_fmaf(xs[at], k2, nm));
      xs[at] = x;
    }
  }
}

__global__ __launch_bounds__(256) void k_fix_sums(const int* __restrict__ rcount, const int* __restrict__ rkey,
                                                  const float* __restrict__ rdelta, const int* __restrict__ ccount,
                                                  const int* __restrict__ ckey, const float* __restrict__ cdelta,
                                                  const float* __restrict__ nm_r, const float* __restrict__ nm_c,
                                                  float* __restrict__ sum_r, float* __restrict__ sum_c, float* __restrict__ nm2_r,
                                                  float* __restrict__ nm2_c, const int* __restrict__ dense_cnt, int Lp, int Sp,
                                                  int slots) {
  const int side = blockIdx.z, b = blockIdx.y;
  const int len = side ? Sp : Lp;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= len || dense_cnt[b] == 0) return;
  const long gl = (long)b * len + idx;
  const int n = min((side ? ccount : rcount)[gl], slots);
  if (n == 0) return;
  const int* key = (side ? ckey : rkey) + gl * slots;
  const float* d = (side ? cdelta : rdelta) + gl * slots;
  float acc = 0.f;
  int last = -1;
  for (int t = 0; t < n; ++t) {           // in index order (the lists are filled in the order the waves arrive)
    int best = 0x7fffffff, at = 0;
    for (int q = 0; q < n; ++q) { const int kq = key[q]; if (kq > last && kq < best) { best = kq; at = q; } }
    acc += d[at];
    last = best;
  }
  float* sum = side ? sum_c : sum_r;
  const float v = sum[gl] + acc;
  sum[gl] = v;
  (side ? nm2_c : nm2_r)[gl] = (side ? nm_c : nm_r)[gl] - __log2f(v);
}

hipError_t launch_exact_lists(const CoarseWs& w, char* base, float inv_ct, const void* feat0, const void* feat1, int in_dtype,
                              int c_in, hipStream_t st) {
  const long groups = (long)w.N * max(w.Lp, w.Sp);
  hipLaunchKernelGGL(k_exact_lists, dim3((unsigned)((groups + 15) / 16), 1, 2), dim3(256), 0, st, feat0, feat1, in_dtype, c_in,
                     (const int*)(base + w.cand_count_b), (const int*)(base + w.cand_j_b), (float*)(base + w.cand_x_b),
                     (float*)(base + w.cand_x), (const int*)(base + w.ccand_count_b), (const int*)(base + w.ccand_i_b),
                     (float*)(base + w.ccand_x_b), (float*)(base + w.ccand_x), (const float*)(base + w.nmr),
                     (const float*)(base + w.nmc), (const int*)(base + w.dense_cnt), w.N, w.L, w.S, w.Lp, w.Sp, w.slots,
                     inv_ct * kLog2e);
  hipLaunchKernelGGL(k_fix_sums, dim3((max(w.Lp, w.Sp) + 255) / 256, w.N, 2), dim3(256), 0, st,
                     (const int*)(base + w.cand_count_b), (const int*)(base + w.cand_j_b), (const float*)(base + w.cand_x),
                     (const int*)(base + w.ccand_count_b), (const int*)(base + w.ccand_i_b), (const float*)(base + w.ccand_x),
                     (const float*)(base + w.nmr), (const float*)(base + w.nmc), (float*)(base + w.rsum), (float*)(base + w.csum),
                     (float*)(base + w.nmr2), (float*)(base + w.nmc2), (const int*)(base + w.dense_cnt), w.Lp, w.Sp, w.slots);
  return hipGetLastError();
}

hipError_t launch_conf_patch(const CoarseWs& w, char* base, float inv_ct, float* conf, hipStream_t st) {
  const long total = (long)w.N * w.Lp * w.slots;
  hipLaunchKernelGGL(k_conf_patch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const int*)(base + w.cand_count),
                     (const int*)(base + w.cand_j), (const float*)(base + w.cand_x), (const int*)(base + w.cand_count_b),
                     (const int*)(base + w.cand_j_b), (const float*)(base + w.cand_x_b), (const float*)(base + w.nmr),
                     (const float*)(base + w.rsum), (const float*)(base + w.nmc), (const float*)(base + w.csum),
                     (const int*)(base + w.dense_cnt), w.N, w.L, w.S, w.Lp, w.Sp, w.slots,
                     inv_ct * kLog2e, conf);
  return hipGetLastError();
}

static int dsm_zsplit(int N, int R) {
  const int wgs = N * ((R + 31) / 32);
  int z = (512 + wgs - 1) / wgs;
  return z < 1 ? 1 : (z > 4 ? 4 : z);
}

}  // namespace fm

using namespace fm;

static bool dsm_shape_ok(int N, int L, int S, int C) { return N > 0 && L > 0 && S > 0 && valid_channels(C); }

extern "C" int fm_dual_softmax_conf_at(const float* feat0, const float* feat1, int N, int L, int S, int C, float temperature,
                                       const float* ofs_r, const float* sum_r, int pitch_r, const float* ofs_c,
                                       const float* sum_c, int pitch_c,
                                       const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, int K, float* conf,
                                       void* stream) {
  if (K == 0) return FM_OK;
  if (!feat0 || !feat1 || !ofs_r || !ofs_c || !sum_r || !sum_c || !b_ids || !i_ids || !j_ids || !conf) return FM_E_NULL;
  if (!(N > 0 && L > 0 && S > 0) || K < 0 || pitch_r < L || pitch_c < S) return FM_E_SHAPE;
  if (!valid_channels(C) || !(temperature > 0.f)) return FM_E_UNSUPPORTED;
  const float k2 = kLog2e / ((float)C * temperature);
  hipLaunchKernelGGL(k_conf_at, dim3((K + 15) / 16), dim3(256), 0, (hipStream_t)stream, feat0, feat1, L, S, C, k2, ofs_r,
                     sum_r, pitch_r, ofs_c, sum_c, pitch_c, b_ids, i_ids, j_ids, K, conf, (float*)nullptr);
  return (int)hipGetLastError();
}

extern "C" size_t fm_dual_softmax_backward_workspace_bytes(int N, int L, int S, int C) {
  if (!dsm_shape_ok(N, L, S, C)) return 0;
  const size_t m = (size_t)(L > S ? L : S);
  return align256((size_t)N * (L + S) * 4) + (size_t)4 * N * m * C * 4;      // u, v + up to 4 partial gradients
}

// the tiled sweep of one side (0: owner = image 0, 1: owner = image 1) in one of its three modes + the combine of its partials
static int dsm_sweep(int mode, int side, const float* feat0, const float* feat1, int N, int L, int S, int C, float k2, float inv_ct,
                     const float* ofs_r, const float* sum_r, int pitch_r, const float* ofs_c, const float* sum_c, int pitch_c,
                     float* v, float* u, float* part, const float* G, float* d_out, hipStream_t st) {
  const float* X = side ? feat1 : feat0;
  const float* Y = side ? feat0 : feat1;
  const int R = side ? S : L, T = side ? L : S;
  const int Z = dsm_zsplit(N, R);
  const dim3 grid((R + 31) / 32, N, Z);
  const int Cp = padded_channels(C);
  const int smem = (64 * (Cp + 4) + 32 * 36 + 32 * 33) * 4;
  hipError_t e = hipSuccess;
#define FM_DSM_LAUNCH(CC, MM)                                                                                              \
  {                                                                                                                        \
    static unsigned long long lds_set = 0;                                                                                 \
    e = ensure_dynamic_lds(&k_dsm_bwd<CC, MM>, (64 * (CC + 4) + 32 * 36 + 32 * 33) * 4, &lds_set);                         \
    if (e != hipSuccess) return (int)e;                                                                                    \
    hipLaunchKernelGGL((k_dsm_bwd<CC, MM>), grid, dim3(256), smem, st, X, Y, R, T, C, side ? ofs_c : ofs_r,                \
                       side ? sum_c : sum_r, side ? pitch_c : pitch_r, side ? ofs_r : ofs_c, side ? sum_r : sum_c,         \
                       side ? pitch_r : pitch_c, side ? u : v, side ? v : u, k2, part, G, side, v, u);                     \
  }
#define FM_DSM_CASE(CC)                                                      \
  case CC:                                                                   \
    if (mode == kDsmSparse) FM_DSM_LAUNCH(CC, kDsmSparse)                    \
    else if (mode == kDsmStats) FM_DSM_LAUNCH(CC, kDsmStats)                 \
    else FM_DSM_LAUNCH(CC, kDsmDense)                                        \
    break;
  switch (Cp) {
    FM_DSM_CASE(64)
    FM_DSM_CASE(128)
    FM_DSM_CASE(256)
    default: return FM_E_UNSUPPORTED;
  }
#undef FM_DSM_CASE
#undef FM_DSM_LAUNCH
  if (mode != kDsmStats) {
    const long n4 = (long)N * R * C / 4;
    hipLaunchKernelGGL(k_dsm_combine, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float4*)part, n4, Z, inv_ct,
                       (float4*)d_out);
  }
  return (int)hipGetLastError();
}

// Backward of the dual softmax for a DENSE dL/dconf (losses/loss.py:44-50, 62-65: the loss terms over all negatives):
// G = dL/dconf [N, L, S] float32 [dev], the softmax statistics as in fm_dual_softmax_backward.  Three tiled sweeps (row /
// column sums of G conf; the two gradients), no [N, L, S] temporary: the workspace is fm_dual_softmax_backward_workspace_bytes.
extern "C" int fm_dual_softmax_backward_dense(const float* feat0, const float* feat1, int N, int L, int S, int C,
                                              float temperature, const float* ofs_r, const float* sum_r, int pitch_r,
                                              const float* ofs_c, const float* sum_c, int pitch_c, const float* G,
                                              void* workspace, size_t workspace_bytes, float* d_feat0, float* d_feat1,
                                              void* stream) {
  if (!feat0 || !feat1 || !ofs_r || !ofs_c || !sum_r || !sum_c || !G || !workspace || !d_feat0 || !d_feat1) return FM_E_NULL;
  if (!(N > 0 && L > 0 && S > 0) || pitch_r < L || pitch_c < S) return FM_E_SHAPE;
  if (!valid_channels(C) || !(temperature > 0.f)) return FM_E_UNSUPPORTED;
  if (workspace_bytes < fm_dual_softmax_backward_workspace_bytes(N, L, S, C) || ((uintptr_t)workspace & 255)) return FM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float inv_ct = 1.0f / ((float)C * temperature), k2 = kLog2e * inv_ct;
  float* v = (float*)workspace;                 // [N][L]  row sums of G conf
  float* u = v + (size_t)N * L;                 // [N][S]  column sums
  float* part = (float*)((char*)workspace + align256((size_t)N * (L + S) * 4));
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)N * (L + S) * 4, st);
  if (e != hipSuccess) return (int)e;
  int r = dsm_sweep(kDsmStats, 0, feat0, feat1, N, L, S, C, k2, inv_ct, ofs_r, sum_r, pitch_r, ofs_c, sum_c, pitch_c, v, u, part, G,
                    nullptr, st);
  if (r != FM_OK) return r;
  r = dsm_sweep(kDsmDense, 0, feat0, feat1, N, L, S, C, k2, inv_ct, ofs_r, sum_r, pitch_r, ofs_c, sum_c, pitch_c, v, u, part, G,
                d_feat0, st);
  if (r != FM_OK) return r;
  return dsm_sweep(kDsmDense, 1, feat0, feat1, N, L, S, C, k2, inv_ct, ofs_r, sum_r, pitch_r, ofs_c, sum_c, pitch_c, v, u, part, G,
                   d_feat1, st);
}

extern "C" int fm_dual_softmax_backward(const float* feat0, const float* feat1, int N, int L, int S, int C, float temperature,
                                        const float* ofs_r, const float* sum_r, int pitch_r, const float* ofs_c,
                                        const float* sum_c, int pitch_c,
                                        const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, const float* gc,
                                        int K, void* workspace, size_t workspace_bytes, float* d_feat0, float* d_feat1,
                                        void* stream) {
  if (!feat0 || !feat1 || !ofs_r || !ofs_c || !sum_r || !sum_c || !workspace || !d_feat0 || !d_feat1) return FM_E_NULL;
  if (K > 0 && (!b_ids || !i_ids || !j_ids || !gc)) return FM_E_NULL;
  if (!(N > 0 && L > 0 && S > 0) || K < 0 || pitch_r < L || pitch_c < S) return FM_E_SHAPE;
  if (!valid_channels(C) || !(temperature > 0.f)) return FM_E_UNSUPPORTED;
  if (workspace_bytes < fm_dual_softmax_backward_workspace_bytes(N, L, S, C) || ((uintptr_t)workspace & 255)) return FM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float inv_ct = 1.0f / ((float)C * temperature), k2 = kLog2e * inv_ct;
  float* v = (float*)workspace;                 // [N][L]  row sums of g c
  float* u = v + (size_t)N * L;                 // [N][S]  column sums
  float* part = (float*)((char*)workspace + align256((size_t)N * (L + S) * 4));
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)N * (L + S) * 4, st);
  if (e != hipSuccess) return (int)e;
  if (K > 0) hipLaunchKernelGGL(k_dsm_uv, dim3((K + 255) / 256), dim3(256), 0, st, b_ids, i_ids, j_ids, gc, K, L, S, v, u);
  for (int side = 0; side < 2; ++side) {
    const int r = dsm_sweep(kDsmSparse, side, feat0, feat1, N, L, S, C, k2, inv_ct, ofs_r, sum_r, pitch_r, ofs_c, sum_c, pitch_c, v,
                            u, part, nullptr, side ? d_feat1 : d_feat0, st);
    if (r != FM_OK) return r;
  }
  if (K > 0)
    hipLaunchKernelGGL(k_dsm_entries, dim3((K + 3) / 4), dim3(256), 0, st, feat0, feat1, L, S, C, b_ids, i_ids, j_ids, gc, K,
                       2.0f * inv_ct, d_feat0, d_feat1);
  return (int)hipGetLastError();
}
