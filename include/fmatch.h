/*
 * fmatch.h - C ABI of libfmatch_hip.so: the MI355X (gfx950) coarse-to-fine matching
 * hot path.  This is the drop-in boundary for the three stage modules the reference
 * calls from network/net.py:75,78,83 (paths relative to the reference repository):
 *
 *   fm_coarse_match    <- CoarseMatching.forward + get_coarse_match
 *                         network/utils/coarse_matching_new.py:43-143
 *   fm_gather_windows  <- FinePreprocess.forward, window crop + select
 *                         network/module/fine_preprocess.py:43-50
 *   fm_fine_match      <- FineMatching.forward
 *                         network/utils/fine_matching_new.py:22-79
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no framework types.  Every pointer
 *     marked [dev] is device memory owned by the caller; the library never
 *     allocates or frees device memory and keeps no global state.
 *   - All work is enqueued on the caller's stream (a hipStream_t passed as void*;
 *     NULL = the default stream); no call synchronises the host except
 *     fm_read_count.  The caller has made the target device current.
 *   - Return value: 0 = FM_OK, < 0 = invalid use (see enum), > 0 = a hipError_t
 *     passed through.  Functions never throw.
 *   - Data-dependent conditions discovered on the device (capacity overflow,
 *     candidate-slot overflow, non-finite / out-of-range descriptors) are
 *     reported through the status word next to the match count (fm_read_count).
 */
#ifndef FMATCH_H_
#define FMATCH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_VERSION 100 /* 0.1.0 */

enum fm_status {
  FM_OK = 0,
  FM_E_NULL = -1,        /* a required pointer is NULL */
  FM_E_SHAPE = -2,       /* L != h0c*w0c, S != h1c*w1c, non-positive size ... */
  FM_E_UNSUPPORTED = -3, /* C > 256 or C % 4 != 0, Cf != 64, W not in {5,7}, thr <= 0 ... */
  FM_E_WORKSPACE = -4,   /* workspace too small / misaligned */
  FM_E_CAPACITY = -5,    /* (device status) more matches than `cap`; M_out = required */
  FM_E_CANDIDATES = -6,  /* (device status) a row or column produced more than cand_slots candidates: call again
                            with mode | FM_MODE_EXACT_SCREENING (then with more cand_slots if it persists) */
  FM_E_RANGE = -7,       /* (device status) a descriptor is not finite or has |x| >= 32768, or the similarities are so
                            large (several thousand: |f0||f1| / (C*temperature)) that the int8 screening margin alone,
                            2^60 in the log2 domain, could overflow the float32 exponentials */
  FM_E_DENSE = -8,       /* (device status) a sample's similarity is flat (more significant entries per 32 x 32 unit
                            than the screening kernel resolves: an untrained network, textureless images); its result
                            is incomplete: call again with mode | FM_MODE_DENSE */
  FM_E_INTERNAL = -9,    /* (device status) the assignment kernel's bounded wait for its predecessor workgroups ran
                            out (never observed; the outputs are incomplete): call again */
  FM_E_STEP = -10        /* (device status) the int8 screening step of an image, estimated from a sample of its rows, is
                            too small for some descriptor outside the sample (an outlier several times larger than the
                            rest, or a sample that fell on textureless cells): what it clipped inflates every screening
                            margin.  The inputs are fine: call again with mode | FM_MODE_EXACT_STEP.  (One step per
                            image bounds the dynamic range the screening can serve: descriptors more than ~5x larger
                            than the rest of their image at C = 256, T = 0.1 still end in FM_E_RANGE - the margin of
                            their coarse codes alone would overflow the exponentials.) */
};

/* `mode` bits of fm_coarse_match / fm_coarse_workspace_bytes_mode (0 = the common path: 4 launches) */
#define FM_MODE_EXACT_SCREENING 1 /* two more kernels re-screen the candidates with the exact softmax denominators */
#define FM_MODE_DENSE 2           /* float16 planes + the dense sum kernel (float32-equivalent product on the matrix
                                     cores) for the samples the screening kernel flags; both exit at once otherwise */
#define FM_MODE_NO_CELL_MAPS 4    /* skip the cell -> match maps of fm_coarse_cell_maps (two returning atomics per match):
                                     for callers that do not use the cell-ordered window crops */
#define FM_MODE_STATS 16          /* also leave the log-softmax offsets of EVERY row and column in the workspace
                                     (fm_coarse_softmax_stats): what fm_dual_softmax_conf_at / _backward read.  No
                                     row or column is skipped as "cannot hold a match"; needs the full-size workspace */
#define FM_MODE_EXACT_STEP 8      /* one more (small) kernel finds the largest |x| of every image first, and the int8
                                     screening step is derived from it instead of from a sample of rows: nothing is
                                     clipped, whatever the data (answers FM_E_STEP) */
#define FM_MODE_FLAT 32           /* a HINT (implies FM_MODE_DENSE): the caller expects flat similarity in every sample
                                     (an untrained network, textureless or occluded scenes - what FM_DEV_ALL_DENSE
                                     reported for the previous call of this kind).  The screening sweep, whose only
                                     finding would be "flat", is skipped: k_prep_split writes the float16 planes
                                     itself, a small kernel forms the stabilisers, and EVERY sample goes to the dense
                                     sum kernel - two launches and one pass over the descriptors fewer.  The hint never
                                     changes what is computed beyond which of the two float32-grade arithmetics serves
                                     a sample (they agree to ~1e-7): peaked data under the hint is correct, only slower */
#define FM_MODE_ALONE 64          /* a HINT about the DEVICE, not the data: this call has the GPU to itself (a single-pair
                                     caller on one stream - the reference's demo/demo.py:95-116).  Launches that cannot
                                     fill the chip then take the grid that is fastest for the kernel alone (two max-pass
                                     workgroups per compute unit, 8-unit screening chunks) instead of the smaller
                                     footprint that lets the kernels of OTHER pairs on other streams run beside them
                                     (the default: DESIGN.md section 5).  Results do not depend on it */

/* element type of the coarse descriptors handed to fm_coarse_match_dtype */
enum fm_dtype { FM_F32 = 0, FM_F16 = 1, FM_BF16 = 2 };

/* device status bits stored in d_count[1] */
#define FM_DEV_CAPACITY 1
#define FM_DEV_CANDIDATES 2
#define FM_DEV_RANGE 4
#define FM_DEV_DENSE 8
#define FM_DEV_INTERNAL 32
#define FM_DEV_STEP 128
#define FM_DEV_ALL_DENSE 256      /* informational, never an error: every sample of the call was served by the dense sum
                                     kernel (fm_read_count_info reports it; FM_MODE_FLAT is the faster way to run the
                                     next call on such data) */

int fm_version(void);
const char* fm_strerror(int status);

/* Candidate slots per coarse row the sparse assignment keeps: conf > thr implies
 * softmax(row) > thr, so at most ceil(1/thr)-1 entries of a row can qualify; the
 * default adds head-room for the float16 screening margin. */
int fm_default_cand_slots(float thr);

/* Bytes of device workspace fm_coarse_match needs (256-byte aligned base): fm_coarse_workspace_bytes for any mode and a
 * conf_matrix request, fm_coarse_workspace_bytes_mode for the given mode bits (want_conf_matrix != 0: as with
 * FM_MODE_DENSE | FM_MODE_EXACT_SCREENING); the common path (mode 0) needs about a quarter of the full size. */
int fm_coarse_workspace_bytes(int N, int L, int S, int C, int cand_slots, size_t* bytes);
int fm_coarse_workspace_bytes_mode(int N, int L, int S, int C, int cand_slots, int mode, int want_conf_matrix,
                                   size_t* bytes);

/*
 * Coarse stage (coarse_matching_new.py:43-143, eval mode).
 *   feat0 [N,L,C], feat1 [N,S,C]     [dev] float32, row-major (l = y*w0c + x)
 *   sim = feat0.feat1^T / (C*temperature); conf = softmax(sim,1)*softmax(sim,2)
 *   keep (b,i,j): conf > thr, both cells >= border_rm from their image border,
 *   conf == row max == column max; emitted sorted by (b,i,j) like torch.where.
 *   scale_px = hw0_i[0]/hw0_c[0] (:126); scale0/scale1 optional [dev] float32 [N,2]
 *   per-sample multipliers (:127-128) or NULL.
 * Outputs (capacity `cap` matches, all [dev]):
 *   b_ids,i_ids,j_ids int64[cap]; mkpts0_c,mkpts1_c float32[cap,2] (x,y px);
 *   mconf float32[cap]; d_count int32[2] = {M, status bits}.
 *   conf_matrix: optional [dev] float32 [N,L,S] (data['conf_matrix'], :70) or NULL.
 *   mode (the former exact_screening flag, same values for 0 / 1): 0 = the common path - prep, int8 max pass,
 *   screening kernel, assignment: candidates are screened in that sweep against lower bounds of the row / column
 *   maxima, which is enough for dual-softmax-trained descriptors; flat similarity reports FM_E_DENSE (a whole sample
 *   has no peaks) or FM_E_CANDIDATES (single rows overflow their cand_slots) through the status word.
 *   FM_MODE_DENSE adds the float16 planes and the dense sum kernel for flagged samples, FM_MODE_EXACT_SCREENING
 *   (implies FM_MODE_DENSE) two more kernels that repeat the screening with the exact softmax denominators, after
 *   which at most 1/thr entries of a row can be candidates; all of them are decided on the device and exit at once
 *   when not needed.  A conf_matrix request implies FM_MODE_DENSE (the sweep reads the float16 planes).  Every entry
 *   of the matrix is within 1e-5 of the float32 reference: entries on a screened sample's lists of significant entries,
 *   and the entries with conf > 0.1 of a sample with flat similarity together with the denominators that contain them,
 *   come from exact float32 dot products; for that the candidate lists of such a sample go down to min(thr, 0.1) - up
 *   to ~10 candidates per row: pass cand_slots >= 16 on flat data (fm_coarse_match_auto does).
 */
int fm_coarse_match(const float* feat0, const float* feat1, int N, int L, int S, int C,
                    int h0c, int w0c, int h1c, int w1c,
                    float temperature, float thr, int border_rm, float scale_px,
                    const float* scale0, const float* scale1,
                    void* workspace, size_t workspace_bytes, int cand_slots, int mode,
                    int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                    float* mkpts0_c, float* mkpts1_c, float* mconf,
                    int cap, int32_t* d_count, float* conf_matrix, void* stream);

/*
 * The same stage for descriptors in float32, float16 or bfloat16 (in_dtype = enum fm_dtype; feat0/feat1 [dev]
 * [N,L,C] / [N,S,C] of that type, 8-byte aligned rows): what a PyTorch-ROCm backbone under autocast hands over,
 * without an up-cast pass in between.  Every half-precision value is exact in float32 and so is the product of
 * two of them, so the result equals fm_coarse_match on the up-cast tensors (the reference's float32 arithmetic on
 * the same numbers, coarse_matching_new.py:64-66).  fm_coarse_match(...) == fm_coarse_match_dtype(..., FM_F32, ...).
 * The workspace does not depend on the input type.
 */
int fm_coarse_match_dtype(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                          int h0c, int w0c, int h1c, int w1c,
                          float temperature, float thr, int border_rm, float scale_px,
                          const float* scale0, const float* scale1,
                          void* workspace, size_t workspace_bytes, int cand_slots, int mode,
                          int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                          float* mkpts0_c, float* mkpts1_c, float* mconf,
                          int cap, int32_t* d_count, float* conf_matrix, void* stream);

/*
 * fm_coarse_match_dtype + a SIDE JOB for callers that go on to fm_fine_match_maps with NCHW float32 fine maps (the
 * reference's layout, network/net.py:56-57): the channels-last copy of image 1's fine map that fm_fine_match_maps makes
 * as its first launch does not depend on the coarse stage at all, and the assignment kernel - 152 small, latency-bound
 * workgroups at 640x480 - leaves the memory system idle.  Here the transpose rides in the assignment kernel's launch as
 * a second workgroup role: one launch fewer per pair and the copy costs no time of its own.
 *   feat_f1 [dev] float32 [Nf, 64, Hf1, Wf1] (NCHW), scratch1 [dev] fm_fine_maps_scratch_bytes(..., layout 0) bytes,
 *   16-byte aligned; afterwards call fm_fine_match_maps with layout = FM_LAYOUT_NCHW_PREPARED and the same scratch.
 * Everything else as fm_coarse_match_dtype (same outputs, same status codes).
 */
int fm_coarse_match_maps(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                         int h0c, int w0c, int h1c, int w1c,
                         float temperature, float thr, int border_rm, float scale_px,
                         const float* scale0, const float* scale1,
                         void* workspace, size_t workspace_bytes, int cand_slots, int mode,
                         int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                         float* mkpts0_c, float* mkpts1_c, float* mconf,
                         int cap, int32_t* d_count, float* conf_matrix,
                         const float* feat_f1, int Nf, int Cf, int Hf1, int Wf1, void* scratch1, void* stream);

/*
 * ONE call for any data: CoarseMatching.forward is one call in the reference (coarse_matching_new.py:43-73), and so is
 * this.  fm_coarse_match / _dtype enqueue exactly the kernels `mode` names and report what the data would have needed
 * through the status word; this function runs the common path, reads that word (a host sync on `stream`, where the
 * reference's torch.where syncs, :109) and answers it by itself:
 *   flat similarity (FM_E_DENSE)          -> the dense sum kernel's part is ADDED to what the common path left in the
 *                                            workspace (float16 planes, dense sums, a second assignment launch: the
 *                                            max pass and the screening are not repeated)
 *   candidate-slot overflow (FM_E_CANDIDATES) -> 16 slots + FM_MODE_EXACT_STEP on dense data, then the exact
 *                                            re-screening (FM_MODE_EXACT_SCREENING), then more slots up to max_cand_slots
 *   clipped int8 step (FM_E_STEP)         -> FM_MODE_EXACT_STEP
 *   the assignment's bounded wait (FM_E_INTERNAL) -> once more
 * Returns FM_OK with *m_out = M (the outputs hold M matches), FM_E_CAPACITY with *m_out = the capacity the outputs need
 * (call again with larger buffers; exact ties can exceed N*min(L,S)), FM_E_RANGE for descriptors that are not finite
 * or out of range, an argument error, or a hipError_t.  Never FM_E_DENSE / FM_E_STEP; FM_E_CANDIDATES only when
 * max_cand_slots slots do not hold a row's candidates even after the exact re-screening (thr < 1 / max_cand_slots).
 *   workspace: fm_coarse_workspace_bytes_auto(N, L, S, C, max_cand_slots) bytes (any mode, max_cand_slots slots;
 *              max_cand_slots = 0 means 64); a smaller workspace is used as far as it goes (FM_E_WORKSPACE beyond).
 *   mode: the options that do not depend on the data (FM_MODE_NO_CELL_MAPS, FM_MODE_STATS); data-dependent bits are
 *         taken as the mode to START with (a caller that knows its data saves the first, failing attempt).
 *   conf_matrix: optional [dev] float32 [N,L,S] or NULL, as in fm_coarse_match.
 *   hint_io: optional HOST int32.  In: 0, or the value a previous call on data of this kind left there - the call
 *            then starts where that one ended (flat data: FM_MODE_FLAT, no screening sweep; ...).  Out: the mode bits
 *            (low byte) and the cand_slots the data asked for beyond the request's own default (second byte, 0 = none)
 *            that served this call, the slot count the serving attempt actually ran with (third byte, always set,
 *            ignored on input: the `cand_slots` to pass to fm_coarse_cell_maps / fm_coarse_softmax_stats for this
 *            workspace), and the number of coarse launch sequences it took (fourth byte, informational).  A hint is never wrong, only possibly slower
 *            than the common path: pass 0 every so often to find out whether the data has changed (the Python layer
 *            does, every 64th call of a shape).
 *   info_out: optional, the raw FM_DEV_* status bits of the attempt that served the call.
 * All other arguments as fm_coarse_match_dtype.
 */
int fm_coarse_workspace_bytes_auto(int N, int L, int S, int C, int max_cand_slots, size_t* bytes);
int fm_coarse_match_auto(const void* feat0, const void* feat1, int in_dtype, int N, int L, int S, int C,
                         int h0c, int w0c, int h1c, int w1c,
                         float temperature, float thr, int border_rm, float scale_px,
                         const float* scale0, const float* scale1,
                         void* workspace, size_t workspace_bytes, int max_cand_slots, int mode,
                         int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                         float* mkpts0_c, float* mkpts1_c, float* mconf,
                         int cap, int32_t* d_count, float* conf_matrix,
                         int32_t* hint_io, int32_t* m_out, int32_t* info_out, void* stream);

/* Copy {M, status} to the host and wait for the stream (the one host sync of the
 * path, where the reference's torch.where syncs: coarse_matching_new.py:109).
 * Returns FM_OK or the error the status bits encode; *m_out is min(M, cap) on
 * success and the required capacity on FM_E_CAPACITY. */
int fm_read_count(const int32_t* d_count, int cap, int32_t* m_out, void* stream);
/* The same, and the raw device status bits (FM_DEV_*, the informational ones included) in *info_out. */
int fm_read_count_info(const int32_t* d_count, int cap, int32_t* m_out, int32_t* info_out, void* stream);

/*
 * Window crop (fine_preprocess.py:43-50): out[m, wy*W+wx, c] =
 * feat_f[b_ids[m], c, stride*y - pad + wy, stride*x - pad + wx] (0 outside the
 * map) with (y,x) = divmod(ids[m], w_c).  layout 0 = NCHW contiguous (the
 * reference's), 1 = NHWC (channels-last storage of the same tensor; with Cf = 64 and
 * W in {5,7} a copy in 16-byte chunks, one wave per window: a window row is W*256
 * contiguous bytes there).
 * The number of windows is min(*d_count, m_max) when d_count != NULL (device
 * side, no host sync), else m_max.  out [m_max, W*W, Cf] float32.
 */
int fm_gather_windows(const float* feat_f, int N, int Cf, int Hf, int Wf, int layout,
                      int W, int stride, int pad, int w_c,
                      const int64_t* b_ids, const int64_t* ids,
                      const int32_t* d_count, int m_max, float* out, void* stream);
/* The same with the element type of the map as an argument (enum fm_dtype): float16 / bfloat16 maps - what a backbone
 * under autocast hands over (network/net.py:56-57) - are read as they are (every value is exact in float32), no up-cast
 * pass over the whole map in front of the crop.  out is float32 either way. */
int fm_gather_windows_dtype(const void* feat_f, int map_dtype, int N, int Cf, int Hf, int Wf, int layout,
                            int W, int stride, int pad, int w_c,
                            const int64_t* b_ids, const int64_t* ids,
                            const int32_t* d_count, int m_max, float* out, void* stream);

/*
 * Cell-ordered window crop for NCHW maps with Cf = 64 and W in {5,7}: one wave per coarse cell in raster
 * order, every XCD a contiguous band of the map (keeps the windows of image 1, whose list order is
 * scattered over the map, inside one L2).  cell_to_match [N, cell_pitch] int32 holds match index + 1 per
 * cell of THIS image (0 = unmatched); ties[0] = number of matches that lost their cell to an exactly
 * tied match, ties[1..] = their indices (at most 1023 listed; beyond that the kernel scans the match
 * list).  fm_coarse_cell_maps returns the maps and tie lists the coarse stage keeps in its workspace
 * (valid until the workspace is reused).  Same outputs as fm_gather_windows.
 * FM_E_UNSUPPORTED when the shape is outside this path: call fm_gather_windows instead.
 */
int fm_coarse_cell_maps(void* workspace, int N, int L, int S, int C, int cand_slots,
                        int32_t** cell0, int* pitch0, int32_t** ties0,
                        int32_t** cell1, int* pitch1, int32_t** ties1);
int fm_gather_windows_cells(const float* feat_f, int N, int Cf, int Hf, int Wf, int W, int stride, int pad,
                            int h_c, int w_c, const int32_t* cell_to_match, int cell_pitch, const int32_t* ties,
                            const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max,
                            float* out, void* stream);

/*
 * Window crop FUSED with the context merge of FinePreprocess (fine_preprocess.py:52-60,
 * fine_concat_coarse_feat = True - the reference's only working setting):
 *     out[m, r, :] = W_w . window[m, r, :] + ctx_bias[b_m, cell_m, :]
 * where merge_feat.weight = [W_w | W_c] ([64, 128]) and
 *     ctx_bias[b, cell, :] = W_c . (down_proj.weight . feat_c[b, cell, :] + down_proj.bias) + merge_feat.bias
 * is a per-cell table [N, h_c*w_c, 64] the caller computes with two plain GEMMs (it does not depend on the
 * window position).  The un-merged windows never reach memory.  packed_w = 16 KiB device buffer filled once
 * per weight update by fm_merge_pack_weights(merge_feat.weight [64,128] row-major).  The product runs as
 * hi/lo-split f16 MFMAs with f32 accumulation (f32-equivalent, ~2^-22 relative); the window operand carries a
 * power-of-two scale that follows every window's largest magnitude (any finite float32 map), the weights a fixed one:
 * |merge_feat.weight| < 16 (a larger weight is packed as NaN: the outputs it touches are NaN, never a wrong number).
 * cell_to_match / ties as in fm_gather_windows_cells, or both NULL for list order.  NCHW, Cf = 64, W in {5,7}.
 */
int fm_merge_pack_weights(const float* merge_w, int Cf, void* packed_w, void* stream);
int fm_gather_merge_windows(const float* feat_f, int N, int Cf, int Hf, int Wf, int W, int stride, int pad,
                            int h_c, int w_c, const int32_t* cell_to_match, int cell_pitch, const int32_t* ties,
                            const void* packed_w, const float* ctx_bias,
                            const int64_t* b_ids, const int64_t* ids, const int32_t* d_count, int m_max,
                            float* out, void* stream);

/*
 * Both images' crops in ONE launch (cell order; what FinePreprocess.forward does in a row,
 * fine_preprocess.py:43-50 / 52-60): out0 from feat_f0 / i_ids, out1 from feat_f1 / j_ids.
 * packed_w NULL = plain crop; non-NULL = crop fused with the context merge (ctx0/ctx1 required).
 * Each crop is bound by one round of memory latency at 640x480, so a second launch only adds ramp and tail.
 */
int fm_gather_windows_pair(const float* feat_f0, const float* feat_f1, int N, int Cf, int Hf0, int Wf0, int Hf1,
                           int Wf1, int W, int stride, int pad, int h0c, int w0c, int h1c, int w1c,
                           const int32_t* cell0, int pitch0, const int32_t* ties0,
                           const int32_t* cell1, int pitch1, const int32_t* ties1,
                           const void* packed_w, const float* ctx0, const float* ctx1,
                           const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids,
                           const int32_t* d_count, int m_max, float* out0, float* out1, void* stream);

/*
 * Fine stage (fine_matching_new.py:50-79): dual-direction window correlation,
 * softmax heat-map, spatial expectation, std.  win0/win1 [m_max, WW, Cf];
 * mix0/mix1 [dev] float32 [WW+1] = Linear(WW,1) weight then bias;
 * out0/out1 [m_max,3] = (x_px, y_px, std) with
 *   xy = mkpts_c + coords*(W/2)*scale_f + W/2        (:75-76)
 */
int fm_fine_match(const float* win0, const float* win1, int m_max, const int32_t* d_count,
                  int WW, int Cf, const float* mix0, const float* mix1,
                  const float* mkpts0_c, const float* mkpts1_c, float scale_f,
                  float* out0, float* out1, void* stream);

/*
 * Window crop + fine stage in one call, straight from the fine maps (fine_preprocess.py:43-50 with the plain
 * windows, then fine_matching_new.py:50-79): for callers without fine-level context layers between the two.  The
 * window tensors never exist: with channels-last maps (layout 1: [N,Hf,Wf,64] storage - a window row is W*256
 * contiguous bytes) every window position of both images is one coalesced 256-byte load of the kernel that does
 * the arithmetic of fm_fine_match.  layout 0 (NCHW, the reference's): the windows of image 0 are read from the
 * NCHW map as it is (the match list walks image 0 in raster order, which the NCHW window loader copes with); image 1,
 * whose windows land wherever the partners are, first gets a channels-last copy in `scratch`
 * (fm_fine_maps_scratch_bytes bytes [dev], 16-byte aligned; 0 bytes / NULL for layout 1) by a tiled transpose.  b_ids / i_ids / j_ids, d_count, mkpts*_c as the coarse stage left them; mix0 / mix1, scale_f,
 * out0 / out1 as in fm_fine_match.  Cf = 64, W in {5,7}.  Results equal fm_gather_windows + fm_fine_match bit for bit.
 */
#define FM_LAYOUT_NCHW_PREPARED 2  /* layout of fm_fine_match_maps*: NCHW maps whose image-1 channels-last copy already sits in
                                     `scratch` - made by fm_coarse_match_maps, whose assignment launch carries the
                                     transpose as a side job (float32 maps) */
size_t fm_fine_maps_scratch_bytes(int N, int Cf, int Hf0, int Wf0, int Hf1, int Wf1, int layout);
int fm_fine_match_maps(const float* feat_f0, const float* feat_f1, int layout, int N, int Cf, int Hf0, int Wf0,
                       int Hf1, int Wf1, int W, int stride, int pad, int w0c, int w1c,
                       const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids,
                       const int32_t* d_count, int m_max, const float* mix0, const float* mix1,
                       const float* mkpts0_c, const float* mkpts1_c, float scale_f,
                       void* scratch, float* out0, float* out1, void* stream);

/*
 * The same with the element type of the maps as an argument (enum fm_dtype): FM_F16 / FM_BF16 maps - what a backbone
 * under autocast hands over (network/net.py:56-57) - are read as they are, no up-cast pass.  Every float16 /
 * bfloat16 value is exact in float32 and the arithmetic is the float32 call's, so the result equals
 * fm_fine_match_maps on the up-cast maps bit for bit.  layout 1: the 2-byte channels-last maps are read in place
 * (128 bytes per pixel).  layout 0 (NCHW): BOTH maps get channels-last copies in `scratch`, in their own element type
 * (half the bytes of the float32 route's copy); fm_fine_maps_scratch_bytes_dtype sizes it.
 */
size_t fm_fine_maps_scratch_bytes_dtype(int N, int Cf, int Hf0, int Wf0, int Hf1, int Wf1, int layout, int map_dtype);
int fm_fine_match_maps_dtype(const void* feat_f0, const void* feat_f1, int map_dtype, int layout, int N, int Cf, int Hf0,
                             int Wf0, int Hf1, int Wf1, int W, int stride, int pad, int w0c, int w1c,
                             const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, const int32_t* d_count,
                             int m_max, const float* mix0, const float* mix1, const float* mkpts0_c,
                             const float* mkpts1_c, float scale_f, void* scratch, float* out0, float* out1, void* stream);

/*
 * Coarse-level context layers in front of the coarse matching (network/net.py:74): the reference's
 * LocalFeatureTransformer (network/module/transformer.py:34-57,78-96, attentions.py:19-46) in its default coarse
 * configuration - d_model 256, 8 heads, linear attention, no masks, any sequence of 'self' / 'cross' layers - on
 * feat0 [dev] float32 [N,L,256] and feat1 [N,S,256].  Float32 arithmetic on the float32 matrix cores; three launches
 * per encoder layer (K/V projection + per-tile KV partials, their fold, the fused query-side layer).
 * packed = fm_coarse_tf_packed_bytes(n_layers) bytes [dev], filled once per weight update by
 * fm_coarse_tf_pack_weights: layers[l] = HOST array of 10 DEVICE pointers in state-dict order - q_proj, k_proj,
 * v_proj, merge .weight [256,256]; mlp.0.weight [512,512]; mlp.2.weight [256,512]; norm1.weight, norm1.bias,
 * norm2.weight, norm2.bias [256].  layer_kinds [host] n_layers ints: 0 = 'self', 1 = 'cross'.  workspace [dev]
 * fm_coarse_tf_workspace_bytes(N, L, S) bytes, 16-byte aligned.  out0 / out1 must not alias the inputs.
 * FM_E_UNSUPPORTED for any other C / nhead / layer kind (the caller keeps its own layers then).
 */
size_t fm_coarse_tf_packed_bytes(int n_layers);
int fm_coarse_tf_workspace_bytes(int N, int L, int S, size_t* bytes);
int fm_coarse_tf_pack_weights(const float* const* const* layers, int n_layers, void* packed, void* stream);
int fm_coarse_transformer(const float* feat0, const float* feat1, int N, int L, int S, int C, int nhead,
                          const int* layer_kinds, int n_layers, const void* packed, void* workspace,
                          size_t workspace_bytes, float* out0, float* out1, void* stream);
/* The same with the reference's padding masks (transformer.py:78-96 mask0 / mask1; attentions.py:35-40: padded query
 * positions get Q = 0, padded source positions K = V = 0; values are still divided by the PADDED length): mask0 [N, L],
 * mask1 [N, S] [dev] bytes (1 = a real token, 0 = padding: torch.bool storage), either may be NULL. */
int fm_coarse_transformer_masked(const float* feat0, const float* feat1, const unsigned char* mask0,
                                 const unsigned char* mask1, int N, int L, int S, int C, int nhead,
                                 const int* layer_kinds, int n_layers, const void* packed, void* workspace,
                                 size_t workspace_bytes, float* out0, float* out1, void* stream);

/*
 * Fine-level context layers between the window crop and the fine matching (network/net.py:79-80): the reference's
 * LocalFeatureTransformer (network/module/transformer.py:34-57,78-96, attentions.py:19-46) in its default fine
 * configuration - d_model 64, 8 heads, layer_names ['self', 'cross'], linear attention, no masks - as ONE kernel,
 * one wave per match working on 32-token slices kept in registers from their load to their store, float32-equivalent
 * products (hi/lo-split float16 MFMAs).  win0/win1, out0/out1 [dev] float32 [m_max, WW, 64], WW in {25, 49}; out may
 * alias win.  packed = fm_fine_tf_packed_bytes() bytes [dev] filled once per weight update by
 * fm_fine_tf_pack_weights(layer0, layer1, ...): each a HOST array of 10 DEVICE pointers in state-dict order -
 * q_proj, k_proj, v_proj, merge .weight [64,64]; mlp.0.weight [128,128]; mlp.2.weight [64,128]; norm1.weight,
 * norm1.bias, norm2.weight, norm2.bias [64] - of layers.0 ('self') and layers.1 ('cross').
 * Operand scales: weights x 2^12 (fixed: |w| < 16, checked by fm_fine_tf_pack_weights), activations x 2^8 and the
 * per-head sums of elu(k)+1 x 2^5 in float16 at the FIRST attempt - window values, their projections and the MLP's
 * hidden layer below 255.9 in magnitude, sum_s (elu(k)+1) of a head feature below 2047 (LayerNorm-ed activations of a
 * trained network are O(1)).  The kernel FOLLOWS the largest magnitude that enters a float16 operand, per match; a
 * match that left the range is recomputed inside the kernel from its input windows with activation (and sum) scales
 * 16x, 256x, 4096x smaller (elements below 2^-3 / scale then lose the lo half of their split: three orders of
 * magnitude below the values that forced the scale down).  Only what does not fit at 2^-4 either (|activation| ~ 1e6,
 * NaN / Inf) or a weight >= 16 is reported: fm_fine_transformer_status ORs FM_DEV_RANGE into *d_status ([dev] int32,
 * zeroed by the caller) - the outputs are then not trustworthy and the caller must use float32 layers for this input
 * (the Python module does).  fm_fine_transformer is the same call without the report.  fm_coarse_transformer has no
 * such limit (its scales follow the data).
 */
size_t fm_fine_tf_packed_bytes(void);
int fm_fine_tf_pack_weights(const float* const* layer0, const float* const* layer1, void* packed, void* stream);
int fm_fine_transformer(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                        const void* packed, float* out0, float* out1, void* stream);
int fm_fine_transformer_status(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                               const void* packed, float* out0, float* out1, int32_t* d_status, void* stream);
/* The same with the FIRST attempt's activation scale chosen by the caller: start_log2_scale in {8, 4, 0, -4} (8 = the
 * default above).  d_lowered ([dev] int32, zeroed by the caller, or NULL) receives by how much the matches of this call
 * went below it (0, 4, 8 or 12 = the largest lowering of any match): a caller that feeds start - lowered back into its
 * next call on similar data (the Python module does) no longer pays for the repeated passes - 0.74 -> 0.45 ms at 3769
 * matches of a random-weight network whose activations leave 2^8 in nearly every workgroup.  A smaller starting scale
 * costs the precision a lowering costs (see above), nothing else. */
int fm_fine_transformer_start(const float* win0, const float* win1, int m_max, const int32_t* d_count, int WW, int Cf,
                              const void* packed, float* out0, float* out1, int32_t* d_status, int start_log2_scale,
                              int32_t* d_lowered, void* stream);

/*
 * Match post-processing (the step after the path; utils/metrics.py:33-81): squared symmetric epipolar distance
 * of every match against a relative pose, and a RANSAC-free inlier score per pair.
 *   mkpts0/mkpts1 [dev] float32 [m_max, kpt_stride] (x, y in pixels first; kpt_stride = 3 for the fine
 *   keypoints [M,3], 2 for plain [M,2]); m_bids [dev] int64 [m_max] (data['m_bids']); the number of matches is
 *   min(*d_count, m_max) when d_count != NULL.  T_0to1 [dev] float32 [N,4,4], K0/K1 [dev] float32 [N,3,3],
 *   row-major.  E = [t]x R (:65-66), points normalised by their intrinsics (:41-42),
 *   d = (p1.E p0)^2 (1/((E p0)_x^2 + (E p0)_y^2) + 1/((E^T p1)_x^2 + (E^T p1)_y^2))   (:47-56).
 * Outputs: epi_errs float32 [m_max] (data['epi_errs']); optional inlier uint8 [m_max] (d < inlier_thr) and
 * per_pair int32 [N,2] = {matches, inliers} per pair, which the caller zeroes beforehand (the kernel adds).
 */
int fm_epipolar_errors(const float* mkpts0, const float* mkpts1, int kpt_stride, const int64_t* m_bids,
                       const int32_t* d_count, int m_max, int N, const float* T_0to1, const float* K0,
                       const float* K1, float inlier_thr, float* epi_errs, unsigned char* inlier,
                       int32_t* per_pair, void* stream);

/*
 * Training surface of the coarse stage without any [N, L, S] array.  The reference's coarse loss
 * (losses/loss.py:27-67 with sparse_spvs, its default) reads data['conf_matrix'] at the supervised entries only.
 *   fm_coarse_softmax_stats   : [dev] pointers into the workspace of a fm_coarse_match call that ran with FM_MODE_STATS
 *                               (or a conf_matrix request): softmax(sim, dim 2)[b,i,j] = exp2(k2 x + nm_r[b*pitch_r + i]) /
 *                               sum_r[b*pitch_r + i], softmax(sim, dim 1)[b,i,j] = exp2(k2 x + nm_c[b*pitch_c + j]) /
 *                               sum_c[b*pitch_c + j], x = feat0[b,i] . feat1[b,j], k2 = log2(e) / (C temperature)
 *                               (stabiliser and denominator kept apart: folded into one offset the float32 rounding
 *                               would cost 1e-5 of a conf near 1).  Valid while the workspace is.
 *   fm_dual_softmax_conf_at   : conf[e] = softmax(sim,1) * softmax(sim,2) at K entries (b_ids, i_ids, j_ids [dev] int64),
 *                               from exact float32 dot products (coarse_matching_new.py:64-68 restricted to the entries).
 *   fm_dual_softmax_backward  : d_feat0 [N,L,C], d_feat1 [N,S,C] = the gradient of sum_e g_e conf_e w.r.t. the
 *                               descriptors, given gc[e] = g_e * conf_e [dev] float32 [K]:
 *                                 dL/dsim_kl = 2 g c [kl supervised] - A_kl u_l - B_kl v_k,  u / v = column / row sums of g c
 *                               (A, B recomputed tile by tile from the statistics; float32 arithmetic).  workspace:
 *                               fm_dual_softmax_backward_workspace_bytes bytes [dev], 256-byte aligned.  feat0 / feat1
 *                               float32.  Entries sharing a row or a column are added with float atomics (the order of
 *                               their additions is the only non-deterministic part).
 */
int fm_coarse_softmax_stats(void* workspace, int N, int L, int S, int C, int cand_slots, const float** nm_r,
                            const float** sum_r, int* pitch_r, const float** nm_c, const float** sum_c, int* pitch_c);
int fm_dual_softmax_conf_at(const float* feat0, const float* feat1, int N, int L, int S, int C, float temperature,
                            const float* nm_r, const float* sum_r, int pitch_r, const float* nm_c, const float* sum_c,
                            int pitch_c, const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, int K,
                            float* conf, void* stream);
size_t fm_dual_softmax_backward_workspace_bytes(int N, int L, int S, int C);
int fm_dual_softmax_backward(const float* feat0, const float* feat1, int N, int L, int S, int C, float temperature,
                             const float* nm_r, const float* sum_r, int pitch_r, const float* nm_c, const float* sum_c,
                             int pitch_c, const int64_t* b_ids, const int64_t* i_ids, const int64_t* j_ids, const float* gc,
                             int K, void* workspace, size_t workspace_bytes, float* d_feat0, float* d_feat1, void* stream);
/* The same for a DENSE dL/dconf: G [dev] float32 [N, L, S] (the reference's loss terms over ALL negatives, losses/loss.py:
 * 44-50, 62-65, hand back a gradient for every entry).  conf is recomputed tile by tile from exact float32 dot products
 * and the statistics; three tiled sweeps (the row / column sums of G conf, then the two gradients) read G three times and
 * write nothing of its size: same workspace as fm_dual_softmax_backward, no [N, L, S] temporary. */
int fm_dual_softmax_backward_dense(const float* feat0, const float* feat1, int N, int L, int S, int C, float temperature,
                                   const float* nm_r, const float* sum_r, int pitch_r, const float* nm_c, const float* sum_c,
                                   int pitch_c, const float* G, void* workspace, size_t workspace_bytes, float* d_feat0,
                                   float* d_feat1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FMATCH_H_ */
