"""`match(img0, img1) -> (kpts0, kpts1, conf)`: the facade the reference exposes as
`net.forward(data)` + `data['mkpts0_f'][:, :2], data['mkpts1_f'][:, :2], data['mconf']`
(network/net.py:40-92, demo/demo.py:108-113).

The feature extractor is NOT part of the accelerated path ("CNN/FPN backbone runs
on PyTorch-ROCm"); `SmallFPN` is a plain torch stand-in with the reference's resolutions (coarse
1/8 with d_model 256, fine 1/2 with d_model 64) so that the pipeline runs end to end with
seeded-random weights.  No checkpoint of the reference exists, so there is no parity claim for it.
Everything after the backbone follows network/net.py:66-83 step by step - coarse context layers, coarse matching,
window crop + context merge, fine context layers, fine matching (`Matcher.forward_features`) - and is checked
against a fixture produced by the reference's own modules (tests: net_tail_small).  In eval mode all five stages are
HIP kernels (the context layers too: transformer.py keeps the torch ops as the trainable definition).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import CoarseMatching, FineMatching, FinePreprocess
from .transformer import LocalFeatureTransformer

DEFAULT_CONFIG = {
    'fine_window_size': 7, 'fine_concat_coarse_feat': True, 'resolution': (8, 2),
    'coarse': {'d_model': 256, 'nhead': 8, 'layer_names': ['self', 'cross'] * 4, 'attention': 'linear'},
    'fine': {'d_model': 64, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear'},
    'match_coarse': {'thr': 0.2, 'border_rm': 2, 'dsmax_temperature': 0.1,
                     'train_coarse_percent': 1.0, 'train_pad_num_gt_min': 200},
}


def _block(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class SmallFPN(nn.Module):
    """1/2 -> 1/4 -> 1/8 encoder with a top-down path back to 1/2 (ResNetFPN_8_2 shaped)."""

    def __init__(self, in_ch=3, d_coarse=256, d_fine=64):
        super().__init__()
        self.l1 = _block(in_ch, 64, 2)
        self.l2 = _block(64, 128, 2)
        self.l3 = _block(128, d_coarse, 2)
        self.up2 = nn.Conv2d(d_coarse, 128, 1)
        self.up1 = nn.Conv2d(128, 64, 1)
        self.out_f = nn.Conv2d(64, d_fine, 3, 1, 1)

    def forward(self, x):
        x1 = self.l1(x)
        x2 = self.l2(x1)
        x3 = self.l3(x2)
        y2 = x2 + F.interpolate(self.up2(x3), scale_factor=2.0, mode='bilinear', align_corners=True)
        y1 = x1 + F.interpolate(self.up1(y2), scale_factor=2.0, mode='bilinear', align_corners=True)
        # zero-mean / unit-variance descriptors per cell (what the LayerNorm of the reference's coarse
        # transformer provides); post-ReLU features are otherwise all-positive and nearly collinear
        x3 = x3 - x3.mean(dim=(2, 3), keepdim=True)      # drop the component common to every cell
        x3 = F.layer_norm(x3.permute(0, 2, 3, 1), (x3.shape[1],)).permute(0, 3, 1, 2).contiguous()
        return x3, self.out_f(y1)


class Matcher(nn.Module):
    """backbone -> coarse context layers -> CoarseMatching -> FinePreprocess -> fine context layers ->
    FineMatching with the reference's `data` dict protocol and sub-module names (network/net.py:24-32,51-92), so
    that a reference state dict addresses the same parameters."""

    def __init__(self, config=None, backbone: nn.Module = None):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config or {})
        self.config = cfg
        self.backbone = backbone or SmallFPN(3, cfg['coarse']['d_model'], cfg['fine']['d_model'])
        self.coarse = LocalFeatureTransformer(cfg['coarse'])
        self.coarse_matching = CoarseMatching(cfg['match_coarse'])
        self.fine_preprocess = FinePreprocess(cfg)
        self.fine = LocalFeatureTransformer(cfg['fine'])
        self.fine_matching = FineMatching(cfg['fine'], window=cfg['fine_window_size'])

    def load_state_dict(self, state_dict, *args, **kwargs):
        """network/net.py:94-102: checkpoints of the reference's Lightning module carry a `matcher.` prefix (and older
        ones `loftr_`); both are stripped.  The reference's sinusoidal position tables (`*.pos_encoding.pe`, dead in
        its forward: net.py:67-68, fine_matching_new.py:20) are dropped if a checkpoint carries them."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith('matcher.'):
                k = k.replace('matcher.', '', 1)
            if k.startswith('loftr_'):
                k = k.replace('loftr_', '', 1)
            if k == 'pos_encoding.pe' or k.endswith('.pos_encoding.pe'):
                continue
            sd[k] = v
        return super().load_state_dict(sd, *args, **kwargs)

    @torch.no_grad()
    def forward(self, data):
        """network/net.py:40-92; results are written into `data` (returned as well, for convenience: the reference
        returns None)"""
        data.update({'bs': data['image0'].size(0),
                     'hw0_i': data['image0'].shape[2:], 'hw1_i': data['image1'].shape[2:]})
        feats_c, feats_f = self.backbone(torch.cat([data['image0'], data['image1']], dim=0))
        (feat_c0, feat_c1), (feat_f0, feat_f1) = feats_c.split(data['bs']), feats_f.split(data['bs'])
        data.update({'hw0_c': feat_c0.shape[2:], 'hw1_c': feat_c1.shape[2:],
                     'hw0_f': feat_f0.shape[2:], 'hw1_f': feat_f1.shape[2:]})
        return self.forward_features(feat_c0, feat_c1, feat_f0, feat_f1, data)

    @torch.no_grad()
    def forward_features(self, feat_c0, feat_c1, feat_f0, feat_f1, data):
        """network/net.py:66-92 on the backbone's maps feat_c* [N,C,hc,wc], feat_f* [N,Cf,Hf,Wf]; `data` holds
        bs, hw0_i, hw1_i."""
        data.update({'hw0_c': feat_c0.shape[2:], 'hw1_c': feat_c1.shape[2:],
                     'hw0_f': feat_f0.shape[2:], 'hw1_f': feat_f1.shape[2:]})
        feat_c0 = feat_c0.flatten(2).transpose(1, 2).contiguous()       # n c h w -> n (h w) c
        feat_c1 = feat_c1.flatten(2).transpose(1, 2).contiguous()
        feat_c0, feat_c1 = self.coarse(feat_c0, feat_c1)                # :74
        self.fine_preprocess.prepare(feat_c0, feat_c1)                  # (match-independent GEMMs of :78, ahead of the sync of :75)
        self.coarse_matching(feat_c0, feat_c1, data)                    # :75
        win0, win1 = self.fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, data)     # :78
        if win0.size(0) != 0:                                           # at least one coarse level predicted
            self.fine.defer_range_check = True                          # (its report is read behind the next launch)
            try:
                win0, win1 = self.fine(win0, win1)                      # :79-80
            finally:
                self.fine.defer_range_check = False
        self.fine_matching(win0, win1, data)                            # :83
        redo = self.fine.resolve_range_check()
        if redo is not None:                                            # the fine kernel could not hold the values: float32 layers
            self.fine_matching(redo[0], redo[1], data)
        data.update({'feat_c0': feat_c0, 'feat_c1': feat_c1, 'feat_f0': feat_f0, 'feat_f1': feat_f1})
        return data

    def match(self, img0: torch.Tensor, img1: torch.Tensor):
        """img0, img1: [N,C,H,W] (or [C,H,W]) float tensors on the GPU, H and W multiples of 8.
        Returns kpts0 [M,2], kpts1 [M,2] (pixels, x then y), conf [M]; batch ids are in
        `self.last['m_bids']`."""
        if img0.dim() == 3:
            img0, img1 = img0[None], img1[None]
        data = self.forward({'image0': img0, 'image1': img1})
        self.last = data
        return data['mkpts0_f'][:, :2], data['mkpts1_f'][:, :2], data['mconf']


def match(img0, img1, matcher: Matcher = None):
    """Functional form with a lazily built default matcher (seeded-random weights)."""
    global _DEFAULT
    if matcher is None:
        if '_DEFAULT' not in globals():
            torch.manual_seed(0)
            _DEFAULT = Matcher().to(img0.device).eval()
        matcher = _DEFAULT
    return matcher.match(img0, img1)
