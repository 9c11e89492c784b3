/*
 * fm_debug.h - diagnostic entry points of libfmatch_hip.so.  NOT part of the drop-in boundary (include/fmatch.h): they
 * launch single kernels of the coarse stage on a workspace a complete call has filled, so that bench.py and the tools/
 * scripts can bracket one kernel with events, and expose the workspace layout to the tests.  A caller of the library
 * never needs them.
 */
#ifndef FM_DEBUG_H_
#define FM_DEBUG_H_

#include "fmatch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic only: workspace layout of fm_coarse_match (40 values: 10 ints, byte offsets, the screening
 * kernel's split geometry, total; order documented in csrc/api.hip) so tests can inspect
 * intermediate statistics. */
int fm_debug_coarse_layout(int N, int L, int S, int C, int cand_slots, int64_t* out, int n_out);

/* Diagnostic only: launch one kernel of the coarse stage (fm_debug_launch_corr: mode 0 = max pass,
 * 1 = dense sum kernel, 2 = exact screening sweep; fm_debug_launch_screen: the screening kernel k_screen / k_thresh + k_screen_rows) on a
 * workspace filled by a previous fm_coarse_match of the same shapes and inputs / zero the candidate counters
 * and scalars so that the sum kernels can run again; used by bench.py to bracket the dominant kernels with
 * events on their own stream.  Modes 1 / 2 and fm_debug_launch_prep_f16 need a full-size workspace
 * (fm_coarse_workspace_bytes). */
int fm_debug_launch_corr(void* workspace, int N, int L, int S, int C, int cand_slots,
                         float temperature, float thr, int mode, void* stream);
int fm_debug_launch_screen(void* workspace, const float* feat0, const float* feat1, int N, int L, int S,
                               int C, int cand_slots, float temperature, float thr, void* stream);   /* float32 rows */
int fm_debug_launch_prep(void* workspace, const float* feat0, const float* feat1, int N, int L, int S, int C,
                         int cand_slots, void* stream);      /* k_prep_split alone; clears the per-call counters */
int fm_debug_launch_prep_f16(void* workspace, const float* feat0, const float* feat1, int N, int L, int S, int C,
                             int cand_slots, int force, void* stream);
int fm_debug_reset_counters(void* workspace, int N, int L, int S, int C, int cand_slots, void* stream);
/* FM_MODE_FLAT's two own launches alone: which = 0 k_prep_split with the float16 planes (clears the per-call counters
 * like fm_debug_launch_prep), which = 1 the stabiliser kernel k_stab.  Full-size workspace. */
int fm_debug_launch_flat(void* workspace, const float* feat0, const float* feat1, int N, int L, int S, int C,
                         int cand_slots, float temperature, float thr, int which, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FM_DEBUG_H_ */
