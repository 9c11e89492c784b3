/*
 * Host-side driver for the C ABI of libfmatch_hip.so (include/fmatch.h): a plain C caller, as the reference's
 * maintainer would write one behind net.forward (network/net.py:75-83).  It dlopens the library, resolves every
 * entry point the header declares and walks the argument checks - every status < 0 in fm_status that is decided on
 * the host - without enqueueing any device work (it runs on a machine without a GPU).  Built with
 * -fsanitize=address against a library whose host objects are instrumented too (make -C featurematching_amd/csrc
 * asan): a stray read or write in the validation / workspace-layout code aborts the run.
 *
 *     abi_driver /path/to/libfmatch_hip_asan.so
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmatch.h"
#include "../../featurematching_amd/csrc/fm_debug.h"

static int checks = 0, failures = 0;
#define EXPECT(expr, want)                                                                  \
  do {                                                                                      \
    const int got_ = (expr);                                                                \
    ++checks;                                                                               \
    if (got_ != (want)) { ++failures; fprintf(stderr, "%s:%d: %s = %d, want %d\n", __FILE__, __LINE__, #expr, got_, (int)(want)); } \
  } while (0)

#define RESOLVE(name) do { *(void**)(&p_##name) = dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)

static __typeof__(fm_version)* p_fm_version;
static __typeof__(fm_strerror)* p_fm_strerror;
static __typeof__(fm_default_cand_slots)* p_fm_default_cand_slots;
static __typeof__(fm_coarse_workspace_bytes)* p_fm_coarse_workspace_bytes;
static __typeof__(fm_coarse_workspace_bytes_mode)* p_fm_coarse_workspace_bytes_mode;
static __typeof__(fm_coarse_match)* p_fm_coarse_match;
static __typeof__(fm_coarse_match_dtype)* p_fm_coarse_match_dtype;
static __typeof__(fm_debug_coarse_layout)* p_fm_debug_coarse_layout;
static __typeof__(fm_debug_launch_corr)* p_fm_debug_launch_corr;
static __typeof__(fm_debug_launch_screen)* p_fm_debug_launch_screen;
static __typeof__(fm_debug_launch_prep_f16)* p_fm_debug_launch_prep_f16;
static __typeof__(fm_debug_launch_prep)* p_fm_debug_launch_prep;
static __typeof__(fm_fine_match_maps)* p_fm_fine_match_maps;
static __typeof__(fm_fine_match_maps_dtype)* p_fm_fine_match_maps_dtype;
static __typeof__(fm_coarse_softmax_stats)* p_fm_coarse_softmax_stats;
static __typeof__(fm_dual_softmax_conf_at)* p_fm_dual_softmax_conf_at;
static __typeof__(fm_dual_softmax_backward)* p_fm_dual_softmax_backward;
static __typeof__(fm_dual_softmax_backward_workspace_bytes)* p_fm_dual_softmax_backward_workspace_bytes;
static __typeof__(fm_dual_softmax_backward_dense)* p_fm_dual_softmax_backward_dense;
static __typeof__(fm_gather_windows_dtype)* p_fm_gather_windows_dtype;
static __typeof__(fm_fine_maps_scratch_bytes_dtype)* p_fm_fine_maps_scratch_bytes_dtype;
static __typeof__(fm_fine_maps_scratch_bytes)* p_fm_fine_maps_scratch_bytes;
static __typeof__(fm_debug_reset_counters)* p_fm_debug_reset_counters;
static __typeof__(fm_coarse_tf_packed_bytes)* p_fm_coarse_tf_packed_bytes;
static __typeof__(fm_coarse_tf_workspace_bytes)* p_fm_coarse_tf_workspace_bytes;
static __typeof__(fm_coarse_tf_pack_weights)* p_fm_coarse_tf_pack_weights;
static __typeof__(fm_coarse_transformer)* p_fm_coarse_transformer;
static __typeof__(fm_read_count)* p_fm_read_count;
static __typeof__(fm_gather_windows)* p_fm_gather_windows;
static __typeof__(fm_coarse_cell_maps)* p_fm_coarse_cell_maps;
static __typeof__(fm_gather_windows_cells)* p_fm_gather_windows_cells;
static __typeof__(fm_merge_pack_weights)* p_fm_merge_pack_weights;
static __typeof__(fm_gather_merge_windows)* p_fm_gather_merge_windows;
static __typeof__(fm_gather_windows_pair)* p_fm_gather_windows_pair;
static __typeof__(fm_fine_match)* p_fm_fine_match;
static __typeof__(fm_epipolar_errors)* p_fm_epipolar_errors;
static __typeof__(fm_coarse_match_maps)* p_fm_coarse_match_maps;
static __typeof__(fm_read_count_info)* p_fm_read_count_info;
static __typeof__(fm_debug_launch_flat)* p_fm_debug_launch_flat;
static __typeof__(fm_fine_transformer_start)* p_fm_fine_transformer_start;
static __typeof__(fm_coarse_match_auto)* p_fm_coarse_match_auto;
static __typeof__(fm_coarse_workspace_bytes_auto)* p_fm_coarse_workspace_bytes_auto;

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s libfmatch_hip.so\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  RESOLVE(fm_version); RESOLVE(fm_strerror); RESOLVE(fm_default_cand_slots); RESOLVE(fm_coarse_workspace_bytes); RESOLVE(fm_coarse_workspace_bytes_mode);
  RESOLVE(fm_coarse_match); RESOLVE(fm_coarse_match_dtype); RESOLVE(fm_debug_coarse_layout); RESOLVE(fm_debug_launch_corr);
  RESOLVE(fm_debug_launch_screen); RESOLVE(fm_debug_launch_prep_f16); RESOLVE(fm_debug_launch_prep); RESOLVE(fm_fine_match_maps); RESOLVE(fm_fine_match_maps_dtype); RESOLVE(fm_coarse_softmax_stats); RESOLVE(fm_dual_softmax_conf_at); RESOLVE(fm_dual_softmax_backward); RESOLVE(fm_dual_softmax_backward_workspace_bytes); RESOLVE(fm_fine_maps_scratch_bytes_dtype); RESOLVE(fm_fine_maps_scratch_bytes); RESOLVE(fm_debug_reset_counters); RESOLVE(fm_read_count);
  RESOLVE(fm_gather_windows); RESOLVE(fm_coarse_cell_maps); RESOLVE(fm_gather_windows_cells);
  RESOLVE(fm_merge_pack_weights); RESOLVE(fm_gather_merge_windows); RESOLVE(fm_gather_windows_pair);
  RESOLVE(fm_fine_match); RESOLVE(fm_epipolar_errors);
  RESOLVE(fm_coarse_tf_packed_bytes); RESOLVE(fm_coarse_tf_workspace_bytes); RESOLVE(fm_coarse_tf_pack_weights);
  RESOLVE(fm_coarse_transformer);
  RESOLVE(fm_coarse_match_maps); RESOLVE(fm_read_count_info); RESOLVE(fm_debug_launch_flat); RESOLVE(fm_fine_transformer_start);
  RESOLVE(fm_coarse_match_auto); RESOLVE(fm_coarse_workspace_bytes_auto);
  RESOLVE(fm_dual_softmax_backward_dense); RESOLVE(fm_gather_windows_dtype);

  EXPECT(p_fm_version(), FM_VERSION);
  for (int s = FM_E_INTERNAL; s <= FM_OK; ++s) EXPECT(p_fm_strerror(s) != NULL && p_fm_strerror(s)[0] != 0, 1);
  EXPECT(strcmp(p_fm_strerror(-99), "unknown fmatch status"), 0);
  EXPECT(p_fm_default_cand_slots(0.2f), 8);
  EXPECT(p_fm_default_cand_slots(0.0f), 64);

  /* workspace query: every shape / configuration refusal */
  size_t bytes = 0;
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 256, 8, &bytes), FM_OK);
  EXPECT(bytes > 10u * 1000 * 1000 && bytes < 80u * 1000 * 1000, 1);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 256, 8, NULL), FM_E_NULL);
  size_t common = 0, full = bytes;
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 0, 0, &common), FM_OK);
  EXPECT(common > 2u * 1000 * 1000 && common <= 8u * 1000 * 1000 && common < full, 1);
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, FM_MODE_DENSE, 0, &bytes), FM_OK);
  EXPECT(bytes == full, 1);
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 0, 1, &bytes), FM_OK);
  EXPECT(bytes == full, 1);
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 128, 0, &bytes), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, FM_MODE_EXACT_STEP, 0, &bytes), FM_OK);
  EXPECT(p_fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 0, 0, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_workspace_bytes(0, 4800, 4800, 256, 8, &bytes), FM_E_SHAPE);
  EXPECT(p_fm_coarse_workspace_bytes(1, -1, 4800, 256, 8, &bytes), FM_E_SHAPE);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 0, 256, 8, &bytes), FM_E_SHAPE);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 258, 8, &bytes), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 512, 8, &bytes), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 256, 2, &bytes), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 256, 12, &bytes), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_workspace_bytes(1, 4800, 4800, 256, 128, &bytes), FM_E_UNSUPPORTED);
  /* ragged / small / large shapes walk the layout arithmetic under the sanitizer */
  for (int n = 1; n <= 64; n *= 4)
    for (int l = 1; l <= 16384; l = l * 3 + 5)
      for (int c = 4; c <= 256; c += 84) EXPECT(p_fm_coarse_workspace_bytes(n, l, 2 * l + 1, c, 8, &bytes), FM_OK);
  int64_t lay[40];
  EXPECT(p_fm_debug_coarse_layout(1, 4800, 4800, 256, 8, lay, 40), FM_OK);
  EXPECT(lay[39] > 0 && lay[4] == 4864 && lay[5] == 4800, 1);
  EXPECT(p_fm_debug_coarse_layout(1, 4800, 4800, 256, 8, lay, 39), FM_E_SHAPE);
  EXPECT(p_fm_debug_coarse_layout(1, 4800, 4800, 256, 8, NULL, 40), FM_E_NULL);
  EXPECT(p_fm_debug_coarse_layout(0, 4800, 4800, 256, 8, lay, 40), FM_E_SHAPE);
  EXPECT(p_fm_debug_coarse_layout(1, 4800, 4800, 256, 3, lay, 40), FM_E_UNSUPPORTED);

  /* coarse stage: refused before any device work */
  void* one = (void*)(uintptr_t)256;     /* a non-NULL, 256-byte aligned address that is never dereferenced */
  void* odd = (void*)(uintptr_t)264;
  int32_t* cnt = (int32_t*)one;
  EXPECT(p_fm_coarse_match(NULL, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, NULL, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, NULL, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, NULL, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match(one, one, 1, 63, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_SHAPE);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 7, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_SHAPE);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, -1, cnt, NULL, NULL), FM_E_SHAPE);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 62, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.0f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 1.0f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.0f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 5, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 16, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_WORKSPACE);
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 128, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);   /* unknown mode bit */
  /* fm_coarse_match_maps: the side job's arguments are checked before anything is enqueued */
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL, 1, 64, 32, 32, one, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, (const float*)one, 1, 64, 32, 32, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, (const float*)one, 1, 32, 32, 32, one, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, (const float*)one, 1, 64, 0, 32, one, NULL), FM_E_SHAPE);
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, (const float*)((char*)one + 4), 1, 64, 32, 32, one, NULL), FM_E_WORKSPACE);
  /* ... and then the coarse call's own (here: thr out of range) */
  EXPECT(p_fm_coarse_match_maps(one, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 1.0f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, (const float*)one, 1, 64, 32, 32, one, NULL), FM_E_UNSUPPORTED);
  {   /* FM_MODE_FLAT is a known mode bit and needs the full-size workspace (the float16 planes) */
    size_t common2 = 0, flatb = 0;
    EXPECT(p_fm_coarse_workspace_bytes_mode(1, 64, 64, 64, 8, 0, 0, &common2), FM_OK);
    EXPECT(p_fm_coarse_workspace_bytes_mode(1, 64, 64, 64, 8, FM_MODE_FLAT, 0, &flatb), FM_OK);
    EXPECT(flatb > common2, 1);
    EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, common2, 8, FM_MODE_FLAT, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_WORKSPACE);
    EXPECT(p_fm_debug_launch_flat(NULL, (const float*)one, (const float*)one, 1, 64, 64, 64, 8, 0.1f, 0.2f, 0, NULL), FM_E_NULL);
    EXPECT(p_fm_debug_launch_flat(one, (const float*)one, (const float*)one, 1, 64, 64, 64, 8, 0.1f, 0.2f, 2, NULL), FM_E_UNSUPPORTED);
    int32_t mm = 0, info = 0;
    EXPECT(p_fm_read_count_info(NULL, 4, &mm, &info, NULL), FM_E_NULL);
  }
  {   /* a workspace sized for the common path is refused when the call needs the dense regions */
    size_t small = 0;
    EXPECT(p_fm_coarse_workspace_bytes_mode(1, 64, 64, 64, 8, 0, 0, &small), FM_OK);
    EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, small, 8, FM_MODE_DENSE, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_WORKSPACE);
    EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, small, 8, 0, one, one, one, one, one, one, 64, cnt, (float*)one, NULL), FM_E_WORKSPACE);
  }
  EXPECT(p_fm_coarse_match(one, one, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, odd, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_WORKSPACE);
  EXPECT(p_fm_coarse_match_dtype(one, one, 7, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_match_dtype(NULL, one, FM_F16, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 8, 0, one, one, one, one, one, one, 64, cnt, NULL, NULL), FM_E_NULL);
  {   /* fm_coarse_match_auto: every refusal comes back before anything is enqueued */
    size_t ab = 0, a16 = 0;
    int32_t am = 0, ainfo = 0, hint = 0;
    EXPECT(p_fm_coarse_workspace_bytes_auto(1, 4800, 4800, 256, 0, &ab), FM_OK);            /* 0 = 64 slots */
    EXPECT(p_fm_coarse_workspace_bytes_auto(1, 4800, 4800, 256, 16, &a16), FM_OK);
    EXPECT(ab > a16 && a16 >= full, 1);
    EXPECT(p_fm_coarse_workspace_bytes_auto(1, 4800, 4800, 256, 12, &ab), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_workspace_bytes_auto(0, 4800, 4800, 256, 0, &ab), FM_E_SHAPE);
    EXPECT(p_fm_coarse_workspace_bytes_auto(1, 4800, 4800, 256, 0, NULL), FM_E_NULL);
#define AUTO(f0, ws, wsb, slots, mode, thr, cntp, mp) \
    p_fm_coarse_match_auto(f0, one, FM_F32, 1, 64, 64, 64, 8, 8, 8, 8, 0.1f, thr, 2, 8.f, NULL, NULL, ws, wsb, slots, mode, \
                           one, one, one, one, one, one, 64, cntp, NULL, &hint, mp, &ainfo, NULL)
    EXPECT(AUTO(one, one, 1u << 30, 0, 0, 0.2f, cnt, NULL), FM_E_NULL);                     /* m_out is required */
    EXPECT(AUTO(NULL, one, 1u << 30, 0, 0, 0.2f, cnt, &am), FM_E_NULL);
    EXPECT(AUTO(one, NULL, 1u << 30, 0, 0, 0.2f, cnt, &am), FM_E_NULL);
    EXPECT(AUTO(one, one, 1u << 30, 0, 0, 0.2f, NULL, &am), FM_E_NULL);
    EXPECT(AUTO(one, one, 1u << 30, 24, 0, 0.2f, cnt, &am), FM_E_UNSUPPORTED);               /* slots not a power of two */
    EXPECT(AUTO(one, one, 1u << 30, 0, 128, 0.2f, cnt, &am), FM_E_UNSUPPORTED);               /* unknown mode bit */
    EXPECT(AUTO(one, one, 1u << 30, 0, 0, 1.0f, cnt, &am), FM_E_UNSUPPORTED);                /* thr */
    EXPECT(AUTO(one, one, 16, 0, 0, 0.2f, cnt, &am), FM_E_WORKSPACE);
    EXPECT(AUTO(one, odd, 1u << 30, 0, 0, 0.2f, cnt, &am), FM_E_WORKSPACE);
    EXPECT(p_fm_coarse_match_auto(one, one, FM_F32, 1, 63, 64, 64, 8, 8, 8, 8, 0.1f, 0.2f, 2, 8.f, NULL, NULL, one, 1u << 30, 0, 0,
                                  one, one, one, one, one, one, 64, cnt, NULL, NULL, &am, NULL, NULL), FM_E_SHAPE);
    EXPECT(hint, 0);                                                                        /* untouched by refusals */
#undef AUTO
  }
  int32_t m = 0;
  EXPECT(p_fm_read_count(NULL, 4, &m, NULL), FM_E_NULL);
  EXPECT(p_fm_read_count(cnt, 4, NULL, NULL), FM_E_NULL);
  int32_t *c0, *c1, *t0, *t1; int q0, q1;
  EXPECT(p_fm_coarse_cell_maps(NULL, 1, 64, 64, 64, 8, &c0, &q0, &t0, &c1, &q1, &t1), FM_E_NULL);
  EXPECT(p_fm_coarse_cell_maps(one, 1, 64, 64, 64, 8, &c0, &q0, &t0, &c1, NULL, &t1), FM_E_NULL);
  EXPECT(p_fm_coarse_cell_maps(one, 0, 64, 64, 64, 8, &c0, &q0, &t0, &c1, &q1, &t1), FM_E_SHAPE);
  EXPECT(p_fm_coarse_cell_maps(one, 1, 64, 64, 64, 6, &c0, &q0, &t0, &c1, &q1, &t1), FM_E_UNSUPPORTED);
  EXPECT(p_fm_coarse_cell_maps(one, 1, 64, 64, 64, 8, &c0, &q0, &t0, &c1, &q1, &t1), FM_OK);
  EXPECT(q0 == 256 && q1 == 64 && c0 != NULL && t1 != NULL, 1);
  EXPECT(p_fm_debug_launch_corr(NULL, 1, 64, 64, 64, 8, 0.1f, 0.2f, 1, NULL), FM_E_NULL);
  EXPECT(p_fm_debug_launch_corr(one, 1, 64, 64, 64, 8, 0.1f, 0.2f, 9, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_debug_launch_corr(one, 0, 64, 64, 64, 8, 0.1f, 0.2f, 1, NULL), FM_E_SHAPE);
  EXPECT(p_fm_debug_launch_screen(one, NULL, one, 1, 64, 64, 64, 8, 0.1f, 0.2f, NULL), FM_E_NULL);
  EXPECT(p_fm_debug_launch_screen(one, one, one, 1, 64, 64, 64, 16 + 1, 0.1f, 0.2f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_debug_launch_prep_f16(one, one, NULL, 1, 64, 64, 64, 8, 1, NULL), FM_E_NULL);
  EXPECT(p_fm_debug_launch_prep(one, NULL, one, 1, 64, 64, 64, 8, NULL), FM_E_NULL);
  EXPECT(p_fm_debug_launch_prep(one, one, one, 1, 64, 64, 63, 8, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_debug_launch_prep_f16(one, one, one, 1, 64, 64, 64, 64 + 1, 1, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_debug_reset_counters(NULL, 1, 64, 64, 64, 8, NULL), FM_E_NULL);
  EXPECT(p_fm_debug_reset_counters(one, 1, 64, 0, 64, 8, NULL), FM_E_SHAPE);

  /* window crop / fine stage */
  float* f = (float*)one; int64_t* ids = (int64_t*)one;
  EXPECT(p_fm_gather_windows(NULL, 1, 64, 8, 8, 0, 7, 4, 2, 2, NULL, NULL, NULL, 0, NULL, NULL), FM_OK);          /* M == 0 */
  EXPECT(p_fm_gather_windows(NULL, 1, 64, 8, 8, 0, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_windows(f, 1, 64, 8, 8, 0, 7, 4, 2, 2, ids, ids, NULL, -4, f, NULL), FM_E_SHAPE);
  EXPECT(p_fm_gather_windows_dtype(f, 5, 1, 64, 8, 8, 0, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_UNSUPPORTED);   /* element type */
  EXPECT(p_fm_gather_windows_dtype(NULL, FM_F16, 1, 64, 8, 8, 0, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_windows_dtype(f, FM_BF16, 1, 64, 8, 8, 1, 7, 4, 2, 2, ids, ids, NULL, 0, f, NULL), FM_OK);         /* M == 0 */
  EXPECT(p_fm_gather_windows(f, 1, 64, 0, 8, 0, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_SHAPE);
  EXPECT(p_fm_gather_windows(f, 1, 64, 8, 8, 2, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_UNSUPPORTED);      /* layout */
  EXPECT(p_fm_gather_windows(f, 1, 64, 8, 8, 0, 17, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_UNSUPPORTED);     /* W > 15 */
  EXPECT(p_fm_gather_windows(f, 1, 62, 8, 8, 1, 7, 4, 2, 2, ids, ids, NULL, 4, f, NULL), FM_E_UNSUPPORTED);      /* NHWC, Cf % 4 */
  int32_t* map = (int32_t*)one;
  EXPECT(p_fm_gather_windows_cells(NULL, 1, 64, 8, 8, 7, 4, 2, 2, 2, NULL, 4, NULL, NULL, NULL, NULL, 0, NULL, NULL), FM_OK);
  EXPECT(p_fm_gather_windows_cells(f, 1, 64, 8, 8, 7, 4, 2, 2, 2, NULL, 4, map, ids, ids, NULL, 3, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_windows_cells(f, 1, 64, 8, 8, 7, 4, 2, 2, 2, map, 3, map, ids, ids, NULL, 3, f, NULL), FM_E_SHAPE);
  EXPECT(p_fm_gather_windows_cells(f, 1, 32, 8, 8, 7, 4, 2, 2, 2, map, 4, map, ids, ids, NULL, 3, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_gather_windows_cells(f, 1, 64, 8, 8, 6, 4, 2, 2, 2, map, 4, map, ids, ids, NULL, 3, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_merge_pack_weights(NULL, 64, one, NULL), FM_E_NULL);
  EXPECT(p_fm_merge_pack_weights(f, 32, one, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_gather_merge_windows(NULL, 1, 64, 8, 8, 7, 4, 2, 2, 2, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL), FM_OK);
  EXPECT(p_fm_gather_merge_windows(f, 1, 64, 8, 8, 7, 4, 2, 2, 2, NULL, 0, NULL, NULL, f, ids, ids, NULL, 3, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_merge_windows(f, 1, 64, 8, 8, 7, 4, 2, 2, 2, map, 4, NULL, one, f, ids, ids, NULL, 3, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_merge_windows(f, 1, 64, 8, 8, 9, 4, 2, 2, 2, NULL, 0, NULL, one, f, ids, ids, NULL, 3, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_gather_windows_pair(f, f, 1, 64, 8, 8, 8, 8, 7, 4, 2, 2, 2, 2, 2, map, 4, map, map, 4, map, NULL, NULL, NULL, ids, ids, ids, NULL, 0, f, f, NULL), FM_OK);
  EXPECT(p_fm_gather_windows_pair(f, NULL, 1, 64, 8, 8, 8, 8, 7, 4, 2, 2, 2, 2, 2, map, 4, map, map, 4, map, NULL, NULL, NULL, ids, ids, ids, NULL, 3, f, f, NULL), FM_E_NULL);
  EXPECT(p_fm_gather_windows_pair(f, f, 1, 64, 8, 8, 8, 8, 7, 4, 2, 2, 2, 2, 2, map, 3, map, map, 4, map, NULL, NULL, NULL, ids, ids, ids, NULL, 3, f, f, NULL), FM_E_SHAPE);
  EXPECT(p_fm_gather_windows_pair(f, f, 1, 32, 8, 8, 8, 8, 7, 4, 2, 2, 2, 2, 2, map, 4, map, map, 4, map, NULL, NULL, NULL, ids, ids, ids, NULL, 3, f, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_match(NULL, NULL, 0, NULL, 49, 64, NULL, NULL, NULL, NULL, 2.f, NULL, NULL, NULL), FM_OK);
  EXPECT(p_fm_fine_match(NULL, f, 3, NULL, 49, 64, f, f, f, f, 2.f, f, f, NULL), FM_E_NULL);
  EXPECT(p_fm_fine_match(f, f, -3, NULL, 49, 64, f, f, f, f, 2.f, f, f, NULL), FM_E_SHAPE);
  /* crop + fine from the maps */
  EXPECT(p_fm_fine_maps_scratch_bytes(2, 64, 240, 320, 240, 320, 0) == (size_t)2 * 64 * 4 * 240 * 320, 1);
  EXPECT(p_fm_fine_maps_scratch_bytes(2, 64, 240, 320, 240, 320, 1) == 0, 1);
  EXPECT(p_fm_fine_match_maps(NULL, NULL, 0, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, 2.f, NULL, NULL, NULL, NULL), FM_OK);   /* M == 0 */
  EXPECT(p_fm_fine_match_maps(f, NULL, 1, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_NULL);
  EXPECT(p_fm_fine_match_maps(f, f, 0, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_NULL);      /* NCHW needs scratch */
  EXPECT(p_fm_fine_match_maps(f, f, 1, 1, 64, 32, 32, 32, 0, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_SHAPE);
  EXPECT(p_fm_fine_match_maps(f, f, 1, 1, 32, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_match_maps(f, f, 1, 1, 64, 32, 32, 32, 32, 9, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_match_maps(f, f, 3, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_UNSUPPORTED);
  /* FM_LAYOUT_NCHW_PREPARED: the scratch a fm_coarse_match_maps call filled is mandatory; float32 maps only */
  EXPECT(p_fm_fine_match_maps(f, f, FM_LAYOUT_NCHW_PREPARED, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_NULL);
  EXPECT(p_fm_fine_match_maps_dtype(f, f, FM_F16, FM_LAYOUT_NCHW_PREPARED, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, (void*)f, f, f, NULL), FM_E_UNSUPPORTED);
  /* fine context layers: the caller's starting scale is one of the kernel's four */
  EXPECT(p_fm_fine_transformer_start(f, f, 4, NULL, 49, 64, (const void*)f, f, f, NULL, 6, NULL, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_transformer_start(f, f, 0, NULL, 49, 64, (const void*)f, f, f, NULL, 4, NULL, NULL), FM_OK);      /* M == 0 */
  EXPECT(p_fm_fine_transformer_start(NULL, f, 4, NULL, 49, 64, (const void*)f, f, f, NULL, 4, NULL, NULL), FM_E_NULL);
  /* training surface: argument checks of the dual-softmax entries */
  {
    const float* pr = NULL; const float* pc = NULL; const float* sr = NULL; const float* sc = NULL; int qr = 0, qc = 0;
    EXPECT(p_fm_coarse_softmax_stats(NULL, 1, 64, 64, 64, 8, &pr, &sr, &qr, &pc, &sc, &qc), FM_E_NULL);
    EXPECT(p_fm_coarse_softmax_stats((void*)f, 1, 64, 64, 64, 8, &pr, &sr, &qr, &pc, &sc, &qc), FM_OK);
    EXPECT(qr, 256); EXPECT(qc, 64);
    EXPECT(p_fm_dual_softmax_conf_at(f, f, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, ids, ids, ids, 0, NULL, NULL), FM_OK);      /* K == 0 */
    EXPECT(p_fm_dual_softmax_conf_at(f, f, 1, 64, 64, 64, 0.1f, f, f, 32, f, f, 64, ids, ids, ids, 4, f, NULL), FM_E_SHAPE);     /* pitch < L */
    EXPECT(p_fm_dual_softmax_conf_at(f, NULL, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, ids, ids, ids, 4, f, NULL), FM_E_NULL);
    EXPECT(p_fm_dual_softmax_conf_at(f, f, 1, 64, 64, 66, 0.1f, f, f, 256, f, f, 64, ids, ids, ids, 4, f, NULL), FM_E_UNSUPPORTED);
    EXPECT((int)p_fm_dual_softmax_backward_workspace_bytes(1, 64, 64, 64), 512 + 4 * 64 * 64 * 4);
    EXPECT((int)p_fm_dual_softmax_backward_workspace_bytes(0, 64, 64, 64), 0);
    EXPECT(p_fm_dual_softmax_backward(f, f, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, ids, ids, ids, f, 4, (void*)f, 16, f, f, NULL), FM_E_WORKSPACE);
    EXPECT(p_fm_dual_softmax_backward(f, f, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, NULL, ids, ids, f, 4, (void*)f, 1u << 30, f, f, NULL), FM_E_NULL);
    /* the dense-gradient form: same workspace, the same refusals */
    EXPECT(p_fm_dual_softmax_backward_dense(f, f, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, f, (void*)f, 16, f, f, NULL), FM_E_WORKSPACE);
    EXPECT(p_fm_dual_softmax_backward_dense(f, f, 1, 64, 64, 64, 0.1f, f, f, 256, f, f, 64, NULL, (void*)f, 1u << 30, f, f, NULL), FM_E_NULL);
    EXPECT(p_fm_dual_softmax_backward_dense(f, f, 1, 64, 64, 64, 0.1f, f, f, 32, f, f, 64, f, (void*)f, 1u << 30, f, f, NULL), FM_E_SHAPE);
    EXPECT(p_fm_dual_softmax_backward_dense(f, f, 1, 64, 64, 62, 0.1f, f, f, 256, f, f, 64, f, (void*)f, 1u << 30, f, f, NULL), FM_E_UNSUPPORTED);
  }
  /* element type of the maps: an unknown one is refused, half-precision NCHW maps need scratch for both copies */
  EXPECT(p_fm_fine_match_maps_dtype(f, f, 7, 1, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_match_maps_dtype(f, f, FM_BF16, 0, 1, 64, 32, 32, 32, 32, 7, 4, 2, 8, 8, ids, ids, ids, NULL, 4, f, f, f, f, 2.f, NULL, f, f, NULL), FM_E_NULL);
  EXPECT((int)p_fm_fine_maps_scratch_bytes_dtype(1, 64, 32, 32, 32, 32, 0, FM_F32), 64 * 4 * 32 * 32);
  EXPECT((int)p_fm_fine_maps_scratch_bytes_dtype(1, 64, 32, 32, 32, 32, 0, FM_F16), 2 * 64 * 2 * 32 * 32);
  EXPECT((int)p_fm_fine_maps_scratch_bytes_dtype(1, 64, 32, 32, 32, 32, 1, FM_F16), 0);
  EXPECT(p_fm_fine_match(f, f, 3, NULL, 36, 64, f, f, f, f, 2.f, f, f, NULL), FM_E_UNSUPPORTED);
  EXPECT(p_fm_fine_match(f, f, 3, NULL, 49, 32, f, f, f, f, 2.f, f, f, NULL), FM_E_UNSUPPORTED);

  EXPECT(p_fm_epipolar_errors(NULL, NULL, 3, NULL, NULL, 0, 1, NULL, NULL, NULL, 1e-4f, NULL, NULL, NULL, NULL), FM_OK);
  EXPECT(p_fm_epipolar_errors(f, f, 3, ids, NULL, 5, 1, f, f, NULL, 1e-4f, f, NULL, NULL, NULL), FM_E_NULL);
  EXPECT(p_fm_epipolar_errors(f, f, 1, ids, NULL, 5, 1, f, f, f, 1e-4f, f, NULL, NULL, NULL), FM_E_SHAPE);
  EXPECT(p_fm_epipolar_errors(f, f, 3, ids, NULL, 5, 0, f, f, f, 1e-4f, f, NULL, NULL, NULL), FM_E_SHAPE);

  {  /* coarse context layers: every host-decided status */
    size_t nb = 0;
    float g[4];
    const int kinds[2] = {0, 1}, bad_kind[1] = {2};
    EXPECT(p_fm_coarse_tf_packed_bytes(0) == 0 ? FM_OK : 1, FM_OK);
    EXPECT(p_fm_coarse_tf_packed_bytes(8) == (size_t)8 * (655360 + 1040) * 4 ? FM_OK : 1, FM_OK);
    EXPECT(p_fm_coarse_tf_workspace_bytes(1, 4800, 4800, NULL), FM_E_NULL);
    EXPECT(p_fm_coarse_tf_workspace_bytes(1, 0, 4800, &nb), FM_E_SHAPE);
    EXPECT(p_fm_coarse_tf_workspace_bytes(1, 4800, 4801, &nb), FM_OK);
    EXPECT(nb == (size_t)(150 + 151 + 2) * 8448 * 4 ? FM_OK : 1, FM_OK);
    EXPECT(p_fm_coarse_tf_pack_weights(NULL, 8, f, NULL), FM_E_NULL);
    EXPECT(p_fm_coarse_tf_pack_weights((const float* const* const*)f, 0, f, NULL), FM_E_UNSUPPORTED);
    /* outputs far from the inputs and from each other (never dereferenced): [1 MiB, +64 KiB) and [2 MiB, +64 KiB) */
    float* o0 = (float*)(uintptr_t)(1u << 20);
    float* o1 = (float*)(uintptr_t)(2u << 20);
    (void)g;
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 8, kinds, 2, NULL, f, nb, o0, o1, NULL), FM_E_NULL);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 0, 256, 8, kinds, 2, f, f, nb, o0, o1, NULL), FM_E_SHAPE);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 128, 8, kinds, 2, f, f, nb, o0, o1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 4, kinds, 2, f, f, nb, o0, o1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 8, bad_kind, 1, f, f, nb, o0, o1, NULL), FM_E_UNSUPPORTED);
    /* aliasing: equal pointers, crossed pointers (feat0 == out1), partially overlapping ranges, overlapping outputs */
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 8, kinds, 2, f, f, nb, (float*)f, o1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(o1, f, 1, 64, 64, 256, 8, kinds, 2, f, f, nb, o0, o1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(f, o0 + 100, 1, 64, 64, 256, 8, kinds, 2, f, f, nb, o0, o1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 8, kinds, 2, f, f, nb, o0, o0 + 64 * 256 - 1, NULL), FM_E_UNSUPPORTED);
    EXPECT(p_fm_coarse_transformer(f, f, 1, 64, 64, 256, 8, kinds, 2, f, f, 16, o0, o1, NULL), FM_E_WORKSPACE);
  }

  printf("abi_driver: %d checks, %d failures\n", checks, failures);
  return failures ? 1 : 0;
}
