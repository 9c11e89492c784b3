"""Drop-in stage modules with the reference's names, signatures and ``data``-dict protocol.

    CoarseMatching.forward(feat_c0, feat_c1, data, mask_c0=None, mask_c1=None) -> None
        (network/utils/coarse_matching_new.py:43)
    FinePreprocess.forward(feat_f0, feat_f1, feat_c0, feat_c1, data) -> (Tensor, Tensor)
        (network/module/fine_preprocess.py:32)
    FineMatching.forward(feat_f0, feat_f1, data) -> None
        (network/utils/fine_matching_new.py:22)

Constructors take the same lower-cased config sub-dicts as network/net.py:29-32.  The
computation runs in the HIP kernels of libfmatch_hip.so (see ops.py); only the learned
``nn.Linear`` layers of FinePreprocess stay on PyTorch-ROCm.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

logger = logging.getLogger("featurematching_amd")


class CoarseMatching(nn.Module):
    """`conf_matrix=True` also materialises the dense data['conf_matrix'] [N,L,S] the reference
    always writes (coarse_matching_new.py:70; only its training loss reads it) - one more sweep and
    N*L*S*4 bytes, so it is opt-in; it is required in training mode, where it carries the gradient of the dual
    softmax back to the descriptors (ops.attach_conf_matrix_grad).
    `gt_pad_sampler=True` selects the older training sampler (network/utils/coarse_matching.py:114-141: predicted
    matches sub-sampled to train_coarse_percent and padded with ground-truth matches up to train_pad_num_gt_min)
    instead of coarse_matching_new.py's plain substitution of the supervision ids (:113-116)."""

    def __init__(self, config, conf_matrix: bool = False, gt_pad_sampler: bool = False):
        super().__init__()
        self.config = config
        self.thr = config['thr']
        self.border_rm = config['border_rm']
        self.train_coarse_percent = config.get('train_coarse_percent', 1.0)
        self.train_pad_num_gt_min = config.get('train_pad_num_gt_min', 200)
        self.temperature = config['dsmax_temperature']
        self.conf_matrix = conf_matrix
        self.gt_pad_sampler = gt_pad_sampler

    def forward(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None):
        """Writes b_ids, i_ids, j_ids, gt_mask, m_bids, mkpts0_c, mkpts1_c, mconf (and conf_matrix on
        request) into ``data`` (coarse_matching_new.py:70,118-141).  mask_c0/mask_c1 are accepted and
        ignored, as in the reference.  In training mode the ids that select the fine windows are the
        supervision ids data['spv_*_ids'] (:113-116); data['conf_matrix'] then carries the gradient of the dual
        softmax (the match lists themselves are not differentiable, as in the reference's @torch.no_grad
        get_coarse_match)."""
        if self.training and not self.conf_matrix:
            raise RuntimeError("training-mode CoarseMatching needs conf_matrix=True (the loss reads data['conf_matrix'])")
        scale = data['hw0_i'][0] / data['hw0_c'][0]
        with torch.no_grad():
            out = ops.coarse_match(feat_c0, feat_c1, data['hw0_c'], data['hw1_c'], scale, self.thr, self.border_rm,
                                   self.temperature, data.get('scale0'), data.get('scale1'),
                                   conf_matrix=self.conf_matrix)
        if self.conf_matrix:
            conf = out['conf_matrix']
            if torch.is_grad_enabled() and (feat_c0.requires_grad or feat_c1.requires_grad):
                conf = ops.attach_conf_matrix_grad(feat_c0, feat_c1, conf, self.temperature, out['_coarse_buffers'])
            data.update({'conf_matrix': conf})
        mconf = out['mconf']
        b_ids, i_ids, j_ids = out['b_ids'], out['i_ids'], out['j_ids']
        mkpts0_c, mkpts1_c = out['mkpts0_c'], out['mkpts1_c']
        if self.training and self.gt_pad_sampler:                            # coarse_matching.py:114-141
            n, l, s = feat_c0.shape[0], feat_c0.shape[1], feat_c1.shape[1]
            num_train = int(n * max(l, s) * self.train_coarse_percent)
            num_pred = b_ids.shape[0]
            assert self.train_pad_num_gt_min < num_train, "min-num-gt-pad should be less than num-train-matches"
            dev = b_ids.device
            if num_pred <= num_train - self.train_pad_num_gt_min:
                pred_idx = torch.arange(num_pred, device=dev)
            else:
                pred_idx = torch.randint(num_pred, (num_train - self.train_pad_num_gt_min,), device=dev)
            gt_idx = torch.randint(len(data['spv_b_ids']), (max(num_train - num_pred, self.train_pad_num_gt_min),), device=dev)
            pick = lambda pred, gt: torch.cat([pred[pred_idx], gt[gt_idx]], dim=0)
            b_ids, i_ids, j_ids = pick(b_ids, data['spv_b_ids']), pick(i_ids, data['spv_i_ids']), pick(j_ids, data['spv_j_ids'])
            mconf = pick(mconf, torch.zeros(len(data['spv_b_ids']), device=dev))    # padded ground truth: mconf == 0
        elif self.training:                                                  # :113-116, :126-134
            b_ids, i_ids, j_ids = data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids']
        if self.training:                                                    # keypoints of the ids in use
            scale0 = scale * data['scale0'][b_ids] if 'scale0' in data else scale
            scale1 = scale * data['scale1'][b_ids] if 'scale1' in data else scale
            w0c, w1c = data['hw0_c'][1], data['hw1_c'][1]
            mkpts0_c = torch.stack([i_ids % w0c, torch.div(i_ids, w0c, rounding_mode='floor')], dim=1) * scale0
            mkpts1_c = torch.stack([j_ids % w1c, torch.div(j_ids, w1c, rounding_mode='floor')], dim=1) * scale1
        if not self.training:     # cell -> match maps for the cell-ordered window crop of FinePreprocess
            data['_fm_coarse'] = out['_coarse_buffers']
        # :137-141.  Only the GT-padding sampler produces entries with mconf == 0; every predicted match has
        # conf > thr > 0, so without it `mconf[mconf != 0]` is mconf itself - and the boolean indexing (a nonzero
        # with its host sync, three launches) is skipped
        padded = self.training and self.gt_pad_sampler
        data.update({'b_ids': b_ids, 'i_ids': i_ids, 'j_ids': j_ids,
                     'gt_mask': mconf == 0, 'm_bids': b_ids,
                     'mkpts0_c': mkpts0_c, 'mkpts1_c': mkpts1_c, 'mconf': mconf[mconf != 0] if padded else mconf})


class FinePreprocess(nn.Module):
    """fused_merge=False keeps crop and context merge apart in eval mode too, cell_ordered=False takes the list-ordered
    crop kernel (tools that time one variant against the other; both default to the fast path)."""

    def __init__(self, config, fused_merge: bool = True, cell_ordered: bool = True):
        super().__init__()
        self.config = config
        self.fused_merge, self.cell_ordered = fused_merge, cell_ordered
        self.cat_c_feat = config['fine_concat_coarse_feat']
        self.W = config['fine_window_size']
        d_model_c = config['coarse']['d_model']
        d_model_f = config['fine']['d_model']
        self.d_model_f = d_model_f
        if self.cat_c_feat:
            self.down_proj = nn.Linear(d_model_c, d_model_f, bias=True)
            self.merge_feat = nn.Linear(2 * d_model_f, d_model_f, bias=True)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")

    def _fused_ok(self, feat_f0, feat_f1, feat_c0, feat_c1, W) -> bool:
        """The fused crop+merge kernel serves eval mode (no autograd through the two Linear layers: inputs that
        require grad under grad mode take the torch layers) on NCHW maps with 64 fine channels and W in {5,7}."""
        wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (feat_f0, feat_f1, feat_c0, feat_c1))
        return self.fused_merge and not self.training and not wants_grad and self.d_model_f == 64 and W in (5, 7) \
            and feat_f0.shape[1] == 64 and feat_f0.is_contiguous() and feat_f1.is_contiguous()

    def _merge_constants(self):
        """(packed W_w fragments, E = W_c . down_proj.weight [64, C], e = W_c . down_proj.bias + merge bias),
        cached until one of the four parameters changes (in-place updates bump torch's version counters)."""
        ps = (self.down_proj.weight, self.down_proj.bias, self.merge_feat.weight, self.merge_feat.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_merge_key', None) != key:
            cf = self.d_model_f
            w_c = self.merge_feat.weight[:, cf:].float()
            self._merge_cache = (ops.pack_merge_weights(self.merge_feat.weight.detach()),
                                 (w_c @ self.down_proj.weight.float()).detach().contiguous(),
                                 (w_c @ self.down_proj.bias.float() + self.merge_feat.bias.float()).detach().contiguous())
            self._merge_key = key
        return self._merge_cache

    def prepare(self, feat_c0, feat_c1):
        """Optional: enqueue the part of forward() that does not depend on the matches - the per-cell context tables
        W_c.(down_proj(feat_c)) + bias, two plain GEMMs - ahead of time.  A caller that runs this BEFORE
        CoarseMatching.forward (whose match count is a host sync) leaves only the crop launch behind the sync
        (matcher.Matcher.forward_features does).  forward() uses the tables when it is handed the same tensors."""
        self._ctx_ready = None
        if not (self.cat_c_feat and self.fused_merge and not self.training and feat_c0.is_cuda):
            return
        if torch.is_grad_enabled() and (feat_c0.requires_grad or feat_c1.requires_grad):
            return
        with torch.no_grad():
            _, e_w, e_b = self._merge_constants()
            self._ctx_ready = ((feat_c0.data_ptr(), feat_c0._version, feat_c1.data_ptr(), feat_c1._version),
                               F.linear(feat_c0.float(), e_w, e_b), F.linear(feat_c1.float(), e_w, e_b))

    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data):
        W = self.W
        stride = data['hw0_f'][0] // data['hw0_c'][0]
        data.update({'W': W})
        b_ids, i_ids, j_ids = data['b_ids'], data['i_ids'], data['j_ids']
        if b_ids.shape[0] == 0:
            feat0 = torch.empty(0, W ** 2, self.d_model_f, device=feat_f0.device)
            feat1 = torch.empty(0, W ** 2, self.d_model_f, device=feat_f0.device)
            return feat0, feat1
        # windows of the matched cells only (the reference unfolds all L cells, then selects)
        # Cell-ordered crop when the ids are this coarse call's own (its cell -> match maps are still in the
        # workspace): every XCD then reads one band of the map.
        cells0 = cells1 = None
        buf = data.get('_fm_coarse') if self.cell_ordered else None
        if buf is not None and buf.b_ids.data_ptr() == b_ids.data_ptr():     # ids are this coarse call's
            cells0, cells1 = buf.cell_maps()
        hw0_c, hw1_c = data['hw0_c'], data['hw1_c']
        if self.cat_c_feat and self._fused_ok(feat_f0, feat_f1, feat_c0, feat_c1, W):
            # inference: crop + merge in one HIP kernel (fm_gather_merge_windows); the un-merged windows never
            # reach memory.  The position-independent half of merge_feat becomes a per-cell table (plain GEMMs).
            with torch.no_grad():
                packed, e_w, e_b = self._merge_constants()
                ready = getattr(self, '_ctx_ready', None)
                self._ctx_ready = None
                if ready is not None and ready[0] == (feat_c0.data_ptr(), feat_c0._version, feat_c1.data_ptr(), feat_c1._version):
                    ctx0, ctx1 = ready[1], ready[2]                 # prepare() ran on these very tensors
                else:
                    ctx0 = F.linear(feat_c0.float(), e_w, e_b)      # [N, L, 64] = W_c.(down_proj(feat_c)) + bias
                    ctx1 = F.linear(feat_c1.float(), e_w, e_b)
                if cells0 is not None and feat_f1.shape[1] == 64:
                    win0, win1 = ops.gather_windows_pair(feat_f0, feat_f1, b_ids, i_ids, j_ids, W, stride, hw0_c, hw1_c,
                                                         (cells0, cells1), packed_w=packed, ctx0=ctx0, ctx1=ctx1)
                else:
                    win0 = ops.gather_merge_windows(feat_f0, packed, ctx0, b_ids, i_ids, W, stride, hw0_c[0], hw0_c[1])
                    win1 = ops.gather_merge_windows(feat_f1, packed, ctx1, b_ids, j_ids, W, stride, hw1_c[0], hw1_c[1])
            return win0, win1
        with torch.no_grad():
            win0 = ops.gather_windows(feat_f0, b_ids, i_ids, W, stride, hw0_c[1], cells=cells0, h_c=hw0_c[0])
            win1 = ops.gather_windows(feat_f1, b_ids, j_ids, W, stride, hw1_c[1], cells=cells1, h_c=hw1_c[0])
        if self.cat_c_feat:      # training (autograd through the two Linear layers) or shapes outside the fused kernel
            feat_c_win = self.down_proj(torch.cat([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0))
            feat_cf_win = self.merge_feat(torch.cat([
                torch.cat([win0, win1], 0),
                feat_c_win[:, None, :].expand(-1, W ** 2, -1)], -1))
            win0, win1 = torch.chunk(feat_cf_win, 2, dim=0)
        return win0, win1


class FineMatching(nn.Module):
    """FineMatching with s2d paradigm; ``window`` fixes the length of the position-mix weights
    (the reference hard-codes Linear(49, 1), i.e. window 7)."""

    def __init__(self, config=None, window: int = 7):
        super().__init__()
        ww = window * window
        self.mix_feat_0 = nn.Linear(ww, 1, bias=True)
        self.mix_feat_1 = nn.Linear(ww, 1, bias=True)

    def _mix(self, lin):
        """[WW + 1] = weights then bias, cached until the layer changes (in-place updates bump the version counters)"""
        key = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version)
        cache = self.__dict__.setdefault('_mix_cache', {})
        hit = cache.get(id(lin))
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]).float().contiguous())
            cache[id(lin)] = hit
        return hit[1]

    @torch.no_grad()
    def forward(self, feat_f0, feat_f1, data):
        M, WW, C = feat_f0.shape
        W = int(math.sqrt(WW))
        scale = data['hw0_i'][0] / data['hw0_f'][0]
        self.M, self.W, self.WW, self.C, self.scale = M, W, WW, C, scale
        if M == 0:
            assert self.training is False, "M is always >0, when training, see coarse_matching.py"
            logger.warning('No matches found in coarse-level.')
            data.update({'expec_f': torch.empty(0, 3, device=feat_f0.device),
                         'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': data['mkpts1_c']})
            return
        k0, k1 = ops.fine_match(feat_f0, feat_f1, self._mix(self.mix_feat_0), self._mix(self.mix_feat_1),
                                data['mkpts0_c'], data['mkpts1_c'], scale)
        data.update({"mkpts0_f": k0, "mkpts1_f": k1})
