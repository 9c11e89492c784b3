#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own modules from /root/reference
(network/utils/coarse_matching_new.py, network/module/fine_preprocess.py,
network/utils/fine_matching_new.py), feeds them the seeded synthetic inputs of
featurematching_amd/synth.py and stores ONLY data: case parameters, (small) inputs
for the adversarial cases and the reference's outputs.  Inputs of the large cases
are regenerated from the portable hash RNG by the tests.

fine_matching_new.py imports loguru and kornia, which are not installed here.  Two
tiny stand-ins are injected into sys.modules: a logger with .warning(), and the two
kornia functions (row a9), restated in oracle/matcher_ref.py from kornia's published
semantics.  Everything else executed is the reference's unmodified code.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from featurematching_amd import synth  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (only for the a9 stand-ins)


def _install_standins():
    loguru = types.ModuleType("loguru")
    loguru.logger = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    sys.modules["loguru"] = loguru
    kornia = types.ModuleType("kornia")
    geometry = types.ModuleType("kornia.geometry")
    subpix = types.ModuleType("kornia.geometry.subpix")
    dsnt = types.ModuleType("kornia.geometry.subpix.dsnt")
    dsnt.spatial_expectation2d = lambda x, normalized_coordinates=True: orc.spatial_expectation2d(x)
    subpix.dsnt = dsnt
    utils = types.ModuleType("kornia.utils")
    grid = types.ModuleType("kornia.utils.grid")
    grid.create_meshgrid = lambda h, w, normalized_coordinates=True, device=None: orc.create_meshgrid(h, w)
    utils.grid = grid
    kornia.geometry, kornia.utils, geometry.subpix = geometry, utils, subpix
    for name, mod in [("kornia", kornia), ("kornia.geometry", geometry), ("kornia.geometry.subpix", subpix),
                      ("kornia.geometry.subpix.dsnt", dsnt), ("kornia.utils", utils), ("kornia.utils.grid", grid)]:
        sys.modules[name] = mod


_install_standins()
from network.utils.coarse_matching_new import CoarseMatching  # noqa: E402
from network.module.fine_preprocess import FinePreprocess  # noqa: E402
from network.utils.fine_matching_new import FineMatching  # noqa: E402

COARSE_CFG = dict(thr=0.2, border_rm=2, train_coarse_percent=1.0, train_pad_num_gt_min=200,
                  dsmax_temperature=0.1)


def ref_coarse(f0, f1, hw0_i, hw1_i, hw0_c, hw1_c, cfg=None, scale0=None, scale1=None):
    cm = CoarseMatching(dict(COARSE_CFG, **(cfg or {}))).eval()
    data = {'hw0_i': hw0_i, 'hw1_i': hw1_i, 'hw0_c': hw0_c, 'hw1_c': hw1_c, 'bs': f0.shape[0]}
    if scale0 is not None:
        data['scale0'], data['scale1'] = torch.as_tensor(scale0), torch.as_tensor(scale1)
    with torch.no_grad():
        cm(torch.as_tensor(f0), torch.as_tensor(f1), data)
    return data


def ref_windows(ff0, ff1, data, w):
    """Reference window cropping: FinePreprocess with the context merge made an
    identity is not expressible, so run its own unfold+select lines through the
    module with cat_c_feat=True and zeroed/identity Linear layers:
    merge_feat = [I | 0] and down_proj = 0 returns exactly the selected windows."""
    cf = ff0.shape[1]
    fp = FinePreprocess({'fine_concat_coarse_feat': True, 'fine_window_size': w,
                         'coarse': {'d_model': 8}, 'fine': {'d_model': cf}}).eval()
    with torch.no_grad():
        fp.down_proj.weight.zero_(); fp.down_proj.bias.zero_()
        fp.merge_feat.weight.zero_(); fp.merge_feat.bias.zero_()
        fp.merge_feat.weight[:, :cf] = torch.eye(cf)
        n = ff0.shape[0]
        l0 = data['hw0_c'][0] * data['hw0_c'][1]
        l1 = data['hw1_c'][0] * data['hw1_c'][1]
        d = dict(data, hw0_f=ff0.shape[2:], hw1_f=ff1.shape[2:])
        w0, w1 = fp(torch.as_tensor(ff0), torch.as_tensor(ff1), torch.zeros(n, l0, 8), torch.zeros(n, l1, 8), d)
    return w0, w1


def ref_fine(win0, win1, data, mix, hw0_f):
    fm = FineMatching({'d_model': win0.shape[-1] if win0.shape[0] else 64}).eval()
    with torch.no_grad():
        fm.mix_feat_0.weight.copy_(torch.as_tensor(mix[0]).view(1, -1)); fm.mix_feat_0.bias.fill_(float(mix[1]))
        fm.mix_feat_1.weight.copy_(torch.as_tensor(mix[2]).view(1, -1)); fm.mix_feat_1.bias.fill_(float(mix[3]))
        d = dict(data, hw0_f=hw0_f)
        fm(win0, win1, d)
    return d['mkpts0_f'], d['mkpts1_f']


def ref_fine_w(win0, win1, data, mix, hw0_f, ww):
    """The reference's FineMatching at another window size: fine_matching_new.py:22-79 derives W from the windows'
    WW (:34); only the two nn.Linear(49, 1) of :18-19 fix 49.  Construct the module as the reference does, swap the
    two layers for nn.Linear(ww, 1) holding the seeded weights, run its UNMODIFIED forward."""
    fm = FineMatching({'d_model': win0.shape[-1] if win0.shape[0] else 64}).eval()
    fm.mix_feat_0 = torch.nn.Linear(ww, 1, bias=True)
    fm.mix_feat_1 = torch.nn.Linear(ww, 1, bias=True)
    with torch.no_grad():
        fm.mix_feat_0.weight.copy_(torch.as_tensor(mix[0]).view(1, -1)); fm.mix_feat_0.bias.fill_(float(mix[1]))
        fm.mix_feat_1.weight.copy_(torch.as_tensor(mix[2]).view(1, -1)); fm.mix_feat_1.bias.fill_(float(mix[3]))
        d = dict(data, hw0_f=hw0_f)
        fm(win0, win1, d)
    return d['mkpts0_f'], d['mkpts1_f']


def w5_case(name, cfgname, dist, w=5):
    """What the metric times (BASELINE.json config 2: 640x480, C = 256, 5x5 fine window) through the reference: its
    CoarseMatching, its FinePreprocess unfold + select at W = 5 (fine_preprocess.py:43-50) and its FineMatching.forward
    with Linear(25, 1) position mixes.  Stored: the coarse outputs, the fine keypoints [M,3], per-window checksums."""
    cfg = dict(synth.CONFIGS[cfgname])
    sh = synth.config_shapes(cfg)
    f0, f1 = synth.coarse_descriptors(cfg['seed'], cfg['n'], sh['l'], cfg['c'], dist)
    hw_i, hw_c = (cfg['h'], cfg['w']), (sh['hc'], sh['wc'])
    data = ref_coarse(f0, f1, hw_i, hw_i, hw_c, hw_c)
    out = pack_coarse(data)
    ff0, ff1 = synth.fine_maps(cfg['seed'], cfg['n'], cfg['cf'], sh['hf'], sh['wf'])
    mix = synth.mix_weights(cfg['seed'], w * w)
    w0, w1 = ref_windows(ff0, ff1, data, w)
    assert w0.shape[1] == w * w
    k0, k1 = ref_fine_w(w0, w1, data, mix, (sh['hf'], sh['wf']), w * w)
    pos = torch.arange(1, w * w + 1, dtype=torch.float64).view(1, w * w, 1)
    ch = torch.arange(1, cfg['cf'] + 1, dtype=torch.float64).view(1, 1, -1)
    out.update(mkpts0_f=k0.numpy(), mkpts1_f=k1.numpy(),
               win0_sum=(w0.double() * pos * ch).sum((1, 2)).numpy(), win1_sum=(w1.double() * pos * ch).sum((1, 2)).numpy(),
               meta=np.array([cfg['n'], cfg['h'], cfg['w'], cfg['c'], cfg['cf'], cfg['seed'], w], np.int64))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: M={out['i_ids'].shape[0]} fine offsets in [{float((k0[:, :2] - data['mkpts0_c']).min()):.2f}, "
          f"{float((k0[:, :2] - data['mkpts0_c']).max()):.2f}] px")


def batch_summary_case(name, cfgname, dist):
    """cfg#3 at its size: ALL samples of the batch through the reference's CoarseMatching, one sample at a time
    (coarse_matching_new.py:43-143 has no cross-sample term), and only a summary per sample stored: M, SHA-256 of the
    (i, j) ids as little-endian int32 pairs in the reference's order, min / max / float64 sum of mconf, float64 sums
    of the coarse keypoints.  Matches whose conf lies within 1e-4 of thr (where float32 re-orderings may flip the
    decision: the tests' guard band) are listed explicitly (`band`: b, i, j + conf) and left OUT of the hashes.  The GPU
    test regenerates the inputs from synth and compares its batch call with it."""
    import hashlib
    cfg = dict(synth.CONFIGS[cfgname])
    sh = synth.config_shapes(cfg)
    hw_i, hw_c = (cfg['h'], cfg['w']), (sh['hc'], sh['wc'])
    ms, shas, cmin, cmax, csum, ksum = [], [], [], [], [], []
    band, band_conf = [], []
    for b in range(cfg['n']):
        f0, f1 = synth.coarse_descriptors(cfg['seed'] + b, 1, sh['l'], cfg['c'], dist)
        d = ref_coarse(f0, f1, hw_i, hw_i, hw_c, hw_c)
        ij = np.stack([d['i_ids'].numpy(), d['j_ids'].numpy()], 1).astype('<i4')
        mc = d['mconf'].numpy()
        inb = np.abs(mc - 0.2) < 1e-4
        for (i, j), cval in zip(ij[inb], mc[inb]):
            band.append((b, int(i), int(j))); band_conf.append(cval)
        ij, mc = ij[~inb], mc[~inb]
        kp0, kp1 = d['mkpts0_c'].numpy()[~inb], d['mkpts1_c'].numpy()[~inb]
        ms.append(ij.shape[0]); shas.append(hashlib.sha256(ij.tobytes()).hexdigest())
        cmin.append(mc.min() if len(mc) else 0.0); cmax.append(mc.max() if len(mc) else 0.0)
        csum.append(mc.astype(np.float64).sum())
        ksum.append([kp0.astype(np.float64).sum(), kp1.astype(np.float64).sum()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), m=np.array(ms, np.int64), sha256=np.array(shas),
                        mconf_min=np.array(cmin, np.float32), mconf_max=np.array(cmax, np.float32),
                        mconf_sum=np.array(csum, np.float64), kpts_sum=np.array(ksum, np.float64),
                        band=np.array(band, np.int32).reshape(-1, 3), band_conf=np.array(band_conf, np.float32),
                        meta=np.array([cfg['n'], cfg['h'], cfg['w'], cfg['c'], cfg['cf'], cfg['seed']], np.int64))
    print(f"{name}: {cfg['n']} samples, M in [{min(ms)}, {max(ms)}] outside the band, total {sum(ms)}; within 1e-4 of thr: {len(band)}")


def pack_coarse(data):
    return dict(b_ids=data['b_ids'].numpy().astype(np.int32), i_ids=data['i_ids'].numpy().astype(np.int32),
                j_ids=data['j_ids'].numpy().astype(np.int32), mconf=data['mconf'].numpy(),
                mkpts0_c=data['mkpts0_c'].numpy().astype(np.float32),
                mkpts1_c=data['mkpts1_c'].numpy().astype(np.float32))


def full_case(name, cfgname, dist, with_fine=True, n=None):
    cfg = dict(synth.CONFIGS[cfgname])
    if n is not None:
        cfg['n'] = n
    sh = synth.config_shapes(cfg)
    f0, f1 = synth.coarse_descriptors(cfg['seed'], cfg['n'], sh['l'], cfg['c'], dist)
    hw_i, hw_c = (cfg['h'], cfg['w']), (sh['hc'], sh['wc'])
    data = ref_coarse(f0, f1, hw_i, hw_i, hw_c, hw_c)
    out = pack_coarse(data)
    out['meta'] = np.array([cfg['n'], cfg['h'], cfg['w'], cfg['c'], cfg['cf'], cfg['seed']], np.int64)
    if with_fine:
        ff0, ff1 = synth.fine_maps(cfg['seed'], cfg['n'], cfg['cf'], sh['hf'], sh['wf'])
        mix = synth.mix_weights(cfg['seed'], 49)
        w0, w1 = ref_windows(ff0, ff1, data, 7)
        k0, k1 = ref_fine(w0, w1, data, mix, (sh['hf'], sh['wf']))
        out['mkpts0_f'], out['mkpts1_f'] = k0.numpy(), k1.numpy()
        # per-window checksums pin the crop geometry without storing 12.5 KB per match
        pos = torch.arange(1, 50, dtype=torch.float64).view(1, 49, 1)
        ch = torch.arange(1, cfg['cf'] + 1, dtype=torch.float64).view(1, 1, -1)
        out['win0_sum'] = (w0.double() * pos * ch).sum((1, 2)).numpy()
        out['win1_sum'] = (w1.double() * pos * ch).sum((1, 2)).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: M={out['i_ids'].shape[0]}")


def merge_case(name, cfgname, dist, w):
    """FinePreprocess as the reference runs it (fine_concat_coarse_feat=True, fine_preprocess.py:52-60) with
    seeded down_proj / merge_feat weights: per-window checksums of the merged windows + the first 3 in full."""
    cfg = dict(synth.CONFIGS[cfgname])
    sh = synth.config_shapes(cfg)
    f0, f1 = synth.coarse_descriptors(cfg['seed'], cfg['n'], sh['l'], cfg['c'], dist)
    hw_i, hw_c = (cfg['h'], cfg['w']), (sh['hc'], sh['wc'])
    data = ref_coarse(f0, f1, hw_i, hw_i, hw_c, hw_c)
    ff0, ff1 = synth.fine_maps(cfg['seed'], cfg['n'], cfg['cf'], sh['hf'], sh['wf'])
    dw, db, mw, mb = synth.merge_weights(cfg['seed'], cfg['c'], cfg['cf'])
    fp = FinePreprocess({'fine_concat_coarse_feat': True, 'fine_window_size': w,
                         'coarse': {'d_model': cfg['c']}, 'fine': {'d_model': cfg['cf']}}).eval()
    with torch.no_grad():
        fp.down_proj.weight.copy_(torch.as_tensor(dw)); fp.down_proj.bias.copy_(torch.as_tensor(db))
        fp.merge_feat.weight.copy_(torch.as_tensor(mw)); fp.merge_feat.bias.copy_(torch.as_tensor(mb))
        d = dict(data, hw0_f=(sh['hf'], sh['wf']), hw1_f=(sh['hf'], sh['wf']))
        m0, m1 = fp(torch.as_tensor(ff0), torch.as_tensor(ff1), torch.as_tensor(f0), torch.as_tensor(f1), d)
    pos = torch.arange(1, w * w + 1, dtype=torch.float64).view(1, w * w, 1)
    ch = torch.arange(1, cfg['cf'] + 1, dtype=torch.float64).view(1, 1, -1)
    out = pack_coarse(data)
    out.update(meta=np.array([cfg['n'], cfg['h'], cfg['w'], cfg['c'], cfg['cf'], cfg['seed'], w], np.int64),
               merged0_sum=(m0.double() * pos * ch).sum((1, 2)).numpy(),
               merged1_sum=(m1.double() * pos * ch).sum((1, 2)).numpy(),
               merged0_head=m0[:3].numpy(), merged1_head=m1[:3].numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: M={out['i_ids'].shape[0]}")


def kat_cases():
    """Small adversarial known-answer cases; inputs are stored with the outputs."""
    cases = {}

    def add(name, f0, f1, hw0_i, hw1_i, hw0_c, hw1_c, cfg=None, scale0=None, scale1=None, fine_seed=None):
        data = ref_coarse(f0, f1, hw0_i, hw1_i, hw0_c, hw1_c, cfg, scale0, scale1)
        d = pack_coarse(data)
        d.update(f0=f0.astype(np.float32), f1=f1.astype(np.float32),
                 hw=np.array([*hw0_i, *hw1_i, *hw0_c, *hw1_c], np.int64),
                 cfg=np.array([(cfg or {}).get('thr', 0.2), (cfg or {}).get('border_rm', 2),
                               (cfg or {}).get('dsmax_temperature', 0.1)], np.float64))
        if scale0 is not None:
            d.update(scale0=np.asarray(scale0, np.float32), scale1=np.asarray(scale1, np.float32))
        if fine_seed is not None:
            n, cf = f0.shape[0], 64
            hf, wf = hw0_c[0] * 4, hw0_c[1] * 4
            ff0, ff1 = synth.fine_maps(fine_seed, n, cf, hf, wf)
            mix = synth.mix_weights(fine_seed, 49)
            w0, w1 = ref_windows(ff0, ff1, data, 7)
            k0, k1 = ref_fine(w0, w1, data, mix, (hf, wf))
            d.update(fine_seed=np.int64(fine_seed), mkpts0_f=k0.numpy(), mkpts1_f=k1.numpy())
        for k, v in d.items():
            cases[f"{name}/{k}"] = v
        print(f"kat {name}: M={d['i_ids'].shape[0]}")

    # exact tie: one descriptor of image 1 duplicated -> both (i, ja), (i, jb) kept (:105-106 uses ==)
    f0, f1 = synth.coarse_descriptors(11, 1, 64, 32, "peaky")
    row = f0[0, 27] + 0.4 * synth.normal(11, 9, (32,))   # interior cell i=27 -> interior slots 28 and 35
    f1[0, 28] = row
    f1[0, 35] = row
    add("tie", f0, f1, (64, 64), (64, 64), (8, 8), (8, 8), fine_seed=11)
    # no match at all: uncorrelated low-magnitude descriptors
    z0 = 0.1 * synth.normal(12, 1, (1, 64, 32)); z1 = 0.1 * synth.normal(12, 2, (1, 64, 32))
    add("empty", z0, z1, (64, 64), (64, 64), (8, 8), (8, 8), fine_seed=12)
    # batch of 3 with different M per sample (sample 1 is uncorrelated)
    f0, f1 = synth.coarse_descriptors(13, 3, 100, 64, "borderline")
    f1[1] = synth.normal(99, 2, (100, 64))
    add("batch3", f0, f1, (80, 80), (80, 80), (10, 10), (10, 10), fine_seed=13)
    # rectangular, L != S, per-sample scale0/scale1 present
    f0 = 3.0 * synth.normal(14, 1, (2, 60, 64))
    f1 = np.concatenate([f0[:, synth.permutation(14, 3, 60)], 3.0 * synth.normal(14, 7, (2, 4, 64))], 1)
    f1 += 0.3 * synth.normal(14, 2, f1.shape)
    s0 = np.array([[1.0, 1.5], [2.0, 0.5]], np.float32); s1 = np.array([[0.75, 1.25], [1.0, 3.0]], np.float32)
    add("rect_scale", f0, f1, (48, 80), (64, 64), (6, 10), (8, 8), scale0=s0, scale1=s1)
    # other threshold / border / temperature
    f0, f1 = synth.coarse_descriptors(15, 1, 144, 64, "borderline")
    add("thr05_b1", f0, f1, (96, 96), (96, 96), (12, 12), (12, 12),
        cfg=dict(thr=0.05, border_rm=1, dsmax_temperature=0.2))
    add("thr0p5_b0", f0, f1, (96, 96), (96, 96), (12, 12), (12, 12), cfg=dict(thr=0.5, border_rm=0))
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **cases)


NET_TAIL = dict(seed=31, n=2, h=128, w=128, c=256, cf=64, nhead=8, layers_c=['self', 'cross'] * 4,
                layers_f=['self', 'cross'], gain=1.0, sigma=1.6)


# the same chain at the size bench.py times forward_features at: ONE 640x480 pair (L = S = 4800, 240x320 fine maps)
NET_TAIL_CFG2 = dict(NET_TAIL, seed=37, n=1, h=480, w=640)


def net_tail_inputs(meta=NET_TAIL):
    """Seeded inputs of the net_tail fixture (feature maps as a backbone would hand them over + all weights), from
    the portable hash RNG: shared with the tests, which regenerate them instead of storing them."""
    seed, n = meta['seed'], meta['n']
    hc, wc, hf, wf = meta['h'] // 8, meta['w'] // 8, meta['h'] // 2, meta['w'] // 2
    l = hc * wc
    f0 = np.empty((n, l, meta['c']), np.float32)
    f1 = np.empty_like(f0)
    for b in range(n):
        z0 = synth.normal(seed + b, 1, (l, meta['c']))
        z1 = synth.normal(seed + b, 2, (l, meta['c']))
        perm = synth.permutation(seed + b, 3, l)
        f0[b] = z0
        f1[b, perm] = z0 + meta['sigma'] / meta['gain'] * z1
    to_map = lambda f: np.ascontiguousarray(f.reshape(n, hc, wc, meta['c']).transpose(0, 3, 1, 2))
    ff0, ff1 = synth.fine_maps(seed, n, meta['cf'], hf, wf)
    w_coarse = synth.transformer_weights(seed + 100, meta['c'], len(meta['layers_c']))
    w_fine = synth.transformer_weights(seed + 200, meta['cf'], len(meta['layers_f']))
    u = lambda st, shape, bound: ((2.0 * synth.uniform(seed + 300, st, int(np.prod(shape))).reshape(shape) - 1.0) * bound).astype(np.float32)
    w_prep = {"down_proj.weight": u(1, (meta['cf'], meta['c']), (6.0 / (meta['cf'] + meta['c'])) ** 0.5),
              "down_proj.bias": u(2, (meta['cf'],), 0.05),
              "merge_feat.weight": u(3, (meta['cf'], 2 * meta['cf']), (6.0 / (3 * meta['cf'])) ** 0.5),
              "merge_feat.bias": u(4, (meta['cf'],), 0.05)}
    mix = synth.mix_weights(seed, 49)
    return dict(feat_c0=to_map(f0), feat_c1=to_map(f1), feat_f0=ff0, feat_f1=ff1, w_coarse=w_coarse, w_fine=w_fine,
                w_prep=w_prep, mix=mix, hw_i=(meta['h'], meta['w']))


def net_tail_case(name="net_tail_small", meta=None):
    """Everything network/net.py:66-83 does after the backbone, run with the REFERENCE's own modules (coarse /
    fine LocalFeatureTransformer, CoarseMatching, FinePreprocess with its context merge, FineMatching) on seeded
    feature maps and weights: the row-a8 fixture."""
    from network.module.transformer import LocalFeatureTransformer
    meta = meta or NET_TAIL
    inp = net_tail_inputs(meta)
    sd = lambda d: {k: torch.as_tensor(v) for k, v in d.items()}
    coarse = LocalFeatureTransformer(dict(d_model=meta['c'], nhead=meta['nhead'], layer_names=meta['layers_c'], attention='linear')).eval()
    fine = LocalFeatureTransformer(dict(d_model=meta['cf'], nhead=meta['nhead'], layer_names=meta['layers_f'], attention='linear')).eval()
    coarse.load_state_dict(sd(inp['w_coarse']))
    fine.load_state_dict(sd(inp['w_fine']))
    fp = FinePreprocess({'fine_concat_coarse_feat': True, 'fine_window_size': 7, 'coarse': {'d_model': meta['c']},
                         'fine': {'d_model': meta['cf']}}).eval()
    fp.load_state_dict(sd(inp['w_prep']))
    cm = CoarseMatching(dict(COARSE_CFG)).eval()
    fm = FineMatching({'d_model': meta['cf']}).eval()
    w0, b0, w1, b1 = inp['mix']
    with torch.no_grad():
        fm.mix_feat_0.weight.copy_(torch.as_tensor(w0).view(1, -1)); fm.mix_feat_0.bias.fill_(float(b0))
        fm.mix_feat_1.weight.copy_(torch.as_tensor(w1).view(1, -1)); fm.mix_feat_1.bias.fill_(float(b1))
        fc0, fc1 = torch.as_tensor(inp['feat_c0']), torch.as_tensor(inp['feat_c1'])
        ff0, ff1 = torch.as_tensor(inp['feat_f0']), torch.as_tensor(inp['feat_f1'])
        data = {'bs': meta['n'], 'hw0_i': inp['hw_i'], 'hw1_i': inp['hw_i'], 'hw0_c': fc0.shape[2:], 'hw1_c': fc1.shape[2:],
                'hw0_f': ff0.shape[2:], 'hw1_f': ff1.shape[2:]}
        c0 = fc0.flatten(2).transpose(1, 2)                    # net.py:69-70 'n c h w -> n (h w) c'
        c1 = fc1.flatten(2).transpose(1, 2)
        c0, c1 = coarse(c0, c1)                                # :74
        cm(c0, c1, data)                                       # :75
        u0, u1 = fp(ff0, ff1, c0, c1, data)                    # :78
        if u0.size(0) != 0:
            u0, u1 = fine(u0, u1)                              # :79-80
        fm(u0, u1, data)                                       # :83
    d = pack_coarse(data)
    d.update(mkpts0_f=data['mkpts0_f'].numpy(), mkpts1_f=data['mkpts1_f'].numpy(),
             c0_sum=c0.double().sum((1, 2)).numpy(), c1_abs=c1.double().abs().sum((1, 2)).numpy(),
             u0_sum=u0.double().sum((1, 2)).numpy(), u1_sum=u1.double().sum((1, 2)).numpy(),
             meta=np.array([meta['seed'], meta['n'], meta['h'], meta['w'], meta['c'], meta['cf']], np.int64))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: M={d['i_ids'].shape[0]} conf range [{d['mconf'].min():.3f}, {d['mconf'].max():.3f}]")


def epipolar_inputs(seed=51, n=3, m=240):
    """Seeded matches / poses / intrinsics of the epi_small fixture (shared with the tests)."""
    u = lambda st, shape: synth.uniform(seed, st, int(np.prod(shape))).reshape(shape)
    b = np.sort((u(1, (m,)) * n).astype(np.int64))
    k0 = np.concatenate([u(2, (m, 2)) * [640, 480], u(3, (m, 1))], 1).astype(np.float32)       # [M,3] like mkpts*_f
    k1 = (k0 + np.concatenate([synth.normal(seed, 4, (m, 2)) * 6.0, np.zeros((m, 1))], 1)).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    for i in range(n):
        w = synth.normal(seed + i, 5, (3,)) * 0.2                      # small rotation (Rodrigues) + translation
        th = float(np.linalg.norm(w)); kx = w / th
        Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
        T[i, :3, :3] = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).astype(np.float32)
        T[i, :3, 3] = synth.normal(seed + i, 6, (3,)).astype(np.float32)
    K = np.tile(np.array([[500.0, 0, 320], [0, 510, 240], [0, 0, 1]], np.float32), (n, 1, 1))
    K0 = K + (u(7, (n, 3, 3)) * [[20, 0, 10], [0, 20, 10], [0, 0, 0]]).astype(np.float32)
    K1 = K + (u(8, (n, 3, 3)) * [[20, 0, 10], [0, 20, 10], [0, 0, 0]]).astype(np.float32)
    return dict(m_bids=b, mkpts0_f=k0, mkpts1_f=k1, T_0to1=T, K0=K0.astype(np.float32), K1=K1.astype(np.float32))


def epipolar_case(name="epi_small"):
    """utils/metrics.py:60-81 compute_symmetrical_epipolar_errors run from the reference's own file.  Its module
    imports cv2, loguru and two kornia helpers that are not installed here: cv2 is not touched by this function
    (empty stand-in); kornia's cross_product_matrix (skew-symmetric matrix of a vector) and
    convert_points_to_homogeneous (append a 1) are restated - third-party, unpinned by the reference."""
    cv2 = types.ModuleType("cv2")
    sys.modules.setdefault("cv2", cv2)
    epi = types.ModuleType("kornia.geometry.epipolar")
    numeric = types.ModuleType("kornia.geometry.epipolar.numeric")

    def cross_product_matrix(x):
        z = torch.zeros_like(x[..., 0])
        return torch.stack([torch.stack([z, -x[..., 2], x[..., 1]], -1), torch.stack([x[..., 2], z, -x[..., 0]], -1),
                            torch.stack([-x[..., 1], x[..., 0], z], -1)], -2)
    numeric.cross_product_matrix = cross_product_matrix
    epi.numeric = numeric
    conv = types.ModuleType("kornia.geometry.conversions")
    conv.convert_points_to_homogeneous = lambda p: torch.cat([p, torch.ones_like(p[..., :1])], -1)
    sys.modules["kornia.geometry.epipolar"] = epi
    sys.modules["kornia.geometry.epipolar.numeric"] = numeric
    sys.modules["kornia.geometry.conversions"] = conv
    sys.modules["kornia.geometry"].epipolar = epi
    sys.modules["kornia.geometry"].conversions = conv
    from utils.metrics import compute_symmetrical_epipolar_errors
    inp = epipolar_inputs()
    data = {k: torch.as_tensor(v) for k, v in inp.items()}
    compute_symmetrical_epipolar_errors(data)
    e = data['epi_errs'].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), epi_errs=e)
    print(f"{name}: M={e.shape[0]} epi range [{e.min():.3e}, {e.max():.3e}], {int((e < 1e-4).sum())} below 1e-4")


def kat_cases_round2():
    """Round-2 known-answer cases (kats_r2.npz): a per-sample scale0/scale1 case with many matches, and a pair
    of cases whose only difference is ONE float32 ulp of one descriptor's scale, chosen by bisection on the
    REFERENCE's own output so that one entry's conf lands just below / just above thr (:99 `conf > thr`)."""
    cases = {}

    def add(name, f0, f1, hw0_i, hw1_i, hw0_c, hw1_c, cfg=None, scale0=None, scale1=None, extra=None):
        data = ref_coarse(f0, f1, hw0_i, hw1_i, hw0_c, hw1_c, cfg, scale0, scale1)
        d = pack_coarse(data)
        d.update(f0=f0.astype(np.float32), f1=f1.astype(np.float32),
                 hw=np.array([*hw0_i, *hw1_i, *hw0_c, *hw1_c], np.int64),
                 cfg=np.array([(cfg or {}).get('thr', 0.2), (cfg or {}).get('border_rm', 2),
                               (cfg or {}).get('dsmax_temperature', 0.1)], np.float64))
        if scale0 is not None:
            d.update(scale0=np.asarray(scale0, np.float32), scale1=np.asarray(scale1, np.float32))
        d.update(extra or {})
        for k, v in d.items():
            cases[f"{name}/{k}"] = v
        print(f"kat {name}: M={d['i_ids'].shape[0]}")
        return data

    # per-sample scales, rectangular maps, L != S, M >= 50
    f0 = 3.0 * synth.normal(21, 1, (2, 12 * 16, 64))
    f1 = np.concatenate([f0[:, synth.permutation(21, 3, 192)], 3.0 * synth.normal(21, 7, (2, 18, 64))], 1)
    f1 += 0.3 * synth.normal(21, 2, f1.shape)
    s0 = np.array([[1.0, 1.5], [2.0, 0.5]], np.float32); s1 = np.array([[0.75, 1.25], [1.0, 3.0]], np.float32)
    add("scale_big", f0, f1, (96, 128), (120, 112), (12, 16), (15, 14), scale0=s0, scale1=s1)

    # conf straddling thr by one ulp of a descriptor scale
    f0, f1 = synth.coarse_descriptors(22, 1, 100, 32, "borderline")
    hw_i, hw_c = (80, 80), (10, 10)
    base = ref_coarse(f0, f1, hw_i, hw_i, hw_c, hw_c)
    ii, jj, cc = base['i_ids'].numpy(), base['j_ids'].numpy(), base['mconf'].numpy()
    pick = int(np.argsort(cc)[len(cc) // 3])             # a mid-confidence interior match
    i_s, j_s = int(ii[pick]), int(jj[pick])
    orig = f1[0, j_s].copy()

    def present(alpha):
        g1 = f1.copy()
        g1[0, j_s] = (np.float32(alpha) * orig).astype(np.float32)
        d = ref_coarse(f0, g1, hw_i, hw_i, hw_c, hw_c)
        hit = (d['i_ids'].numpy() == i_s) & (d['j_ids'].numpy() == j_s)
        return bool(hit.any()), g1, d

    lo, hi = np.float32(0.0), np.float32(1.0)            # alpha = 0 kills the match, alpha = 1 keeps it
    assert not present(lo)[0] and present(hi)[0]
    while np.nextafter(lo, np.float32(2.0), dtype=np.float32) < hi:
        mid = np.float32((np.float64(lo) + np.float64(hi)) / 2)
        if present(mid)[0]:
            hi = mid
        else:
            lo = mid
    _, g_lo, _ = present(lo)
    _, g_hi, d_hi = present(hi)
    hit = (d_hi['i_ids'].numpy() == i_s) & (d_hi['j_ids'].numpy() == j_s)
    conf_hi = float(d_hi['mconf'].numpy()[hit][0])
    print(f"thr straddle: entry ({i_s},{j_s}) alpha {lo!r} -> absent, {hi!r} -> present with conf {conf_hi!r}")
    meta = dict(straddle=np.array([i_s, j_s], np.int64), straddle_conf=np.float64(conf_hi))
    add("thr_below", f0, g_lo, hw_i, hw_i, hw_c, hw_c, extra=meta)
    add("thr_above", f0, g_hi, hw_i, hw_i, hw_c, hw_c, extra=meta)
    np.savez_compressed(os.path.join(HERE, "kats_r2.npz"), **cases)


def masked_coarse_transformer_case(name="tf_masked_coarse"):
    """The reference's LocalFeatureTransformer in its COARSE configuration (d_model 256, 8 heads, linear attention) WITH
    padding masks (transformer.py:78-96, attentions.py:35-40): N = 2, L = 77, S = 130, ['self', 'cross'] x 2, the tails of
    the samples masked out (one sample of image 1 unmasked).  Inputs and weights from the portable RNG; stored: the outputs."""
    from network.module.transformer import LocalFeatureTransformer
    seed, n, l, s_, d = 43, 2, 77, 130, 256
    names = ['self', 'cross', 'self', 'cross']
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=names, attention='linear')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, len(names)).items()})
    x0 = torch.as_tensor((2.0 * synth.normal(seed, 1, (n, l, d))).astype(np.float32))
    x1 = torch.as_tensor((2.0 * synth.normal(seed, 2, (n, s_, d))).astype(np.float32))
    m0 = torch.ones(n, l, dtype=torch.bool); m0[0, 60:] = False; m0[1, 33:] = False
    m1 = torch.ones(n, s_, dtype=torch.bool); m1[0, 100:] = False
    with torch.no_grad():
        y0, y1 = tf(x0, x1, m0, m1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), out0=y0.numpy(), out1=y1.numpy(),
                        mask0=m0.numpy(), mask1=m1.numpy(), meta=np.array([seed, n, l, s_, d], np.int64))
    print(f"{name}: out0 {tuple(y0.shape)} |max| {float(y0.abs().max()):.3f}")


def full_attention_case(name="tf_full_small"):
    """The reference's LocalFeatureTransformer with attention='full' (attentions.py:54-79), with and without padding
    masks, on the inputs and weights of tf_masked_small (d_model 64, 8 heads, ['self', 'cross'], N = 2, L = 40, S = 36)."""
    from network.module.transformer import LocalFeatureTransformer
    seed, n, l, s_, d = 41, 2, 40, 36, 64
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=['self', 'cross'], attention='full')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, 2).items()})
    x0 = torch.as_tensor(synth.normal(seed, 1, (n, l, d)))
    x1 = torch.as_tensor(synth.normal(seed, 2, (n, s_, d)))
    m0 = torch.ones(n, l, dtype=torch.bool); m0[0, 33:] = False; m0[1, 25:] = False
    m1 = torch.ones(n, s_, dtype=torch.bool); m1[0, 30:] = False
    with torch.no_grad():
        y0, y1 = tf(x0, x1)
        z0, z1 = tf(x0, x1, m0, m1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), out0=y0.numpy(), out1=y1.numpy(), mout0=z0.numpy(), mout1=z1.numpy(),
                        mask0=m0.numpy(), mask1=m1.numpy(), meta=np.array([seed, n, l, s_, d], np.int64))
    print(f"{name}: out0 {tuple(y0.shape)} |max| {float(y0.abs().max()):.3f}, masked rows NaN: {bool(torch.isnan(z0).any())}")


def masked_transformer_case(name="tf_masked_small"):
    """The reference's LocalFeatureTransformer WITH padding masks (network/module/transformer.py:78-96,
    attentions.py:35-40) on a small seeded problem: d_model 64, 8 heads, ['self', 'cross'], N = 2, L = 40, S = 36, the
    last positions of each sample masked out.  Inputs and weights come from the portable RNG (the test regenerates
    them); stored: the outputs."""
    from network.module.transformer import LocalFeatureTransformer
    seed, n, l, s_, d = 41, 2, 40, 36, 64
    names = ['self', 'cross']
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=names, attention='linear')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, 2).items()})
    x0 = torch.as_tensor(synth.normal(seed, 1, (n, l, d)))
    x1 = torch.as_tensor(synth.normal(seed, 2, (n, s_, d)))
    m0 = torch.ones(n, l, dtype=torch.bool); m0[0, 33:] = False; m0[1, 25:] = False
    m1 = torch.ones(n, s_, dtype=torch.bool); m1[0, 30:] = False
    with torch.no_grad():
        y0, y1 = tf(x0, x1, m0, m1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), out0=y0.numpy(), out1=y1.numpy(),
                        mask0=m0.numpy(), mask1=m1.numpy(), meta=np.array([seed, n, l, s_, d], np.int64))
    print(f"{name}: out0 {tuple(y0.shape)} |max| {float(y0.abs().max()):.3f}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "r4":     # the fixtures added in round 4 only
        masked_transformer_case()
        full_case("l9600_peaky", "l9600", "peaky")
        full_case("l9600_borderline", "l9600", "borderline", with_fine=False)
        full_case("cfg5_peaky", "cfg5", "peaky")         # now with the fine stage (W = 7)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r2":       # only the round-2 cases (the others are unchanged)
        kat_cases_round2()
        net_tail_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "net_tail":
        net_tail_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r6":     # the fixtures added in round 6 only
        w5_case("cfg2_peaky_w5", "cfg2", "peaky")
        w5_case("cfg2_borderline_w5", "cfg2", "borderline")
        batch_summary_case("cfg3_all64_peaky", "cfg3", "peaky")
        batch_summary_case("cfg3_all64_borderline", "cfg3", "borderline")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r5":     # the fixtures added in round 5 only
        net_tail_case("net_tail_cfg2", NET_TAIL_CFG2)
        full_attention_case()
        masked_coarse_transformer_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "epi":
        epipolar_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r3":     # the fixtures added in round 3 only
        full_case("cfg2_mixed", "cfg2", "mixed")
        full_case("cfg3_first2_borderline", "cfg3", "borderline", with_fine=False, n=2)
        full_case("cfg5_borderline", "cfg5", "borderline", with_fine=False)
        sys.exit(0)
    kat_cases()
    kat_cases_round2()
    net_tail_case()
    epipolar_case()
    full_case("cfg1_peaky", "cfg1", "peaky")
    full_case("cfg1_borderline", "cfg1", "borderline")
    full_case("cfg2_peaky", "cfg2", "peaky")
    full_case("cfg2_borderline", "cfg2", "borderline")
    full_case("cfg3_first2_peaky", "cfg3", "peaky", n=2)
    full_case("cfg5_peaky", "cfg5", "peaky")
    merge_case("merge_cfg1_w7", "cfg1", "peaky", 7)
    merge_case("merge_cfg2_w5", "cfg2", "borderline", 5)
    # round 3: the reference's dual softmax at S = 16384 and at batch size on data whose conf values are NOT all 1.0,
    # and peaked data with textureless cells (flat rows / columns next to peaked ones)
    full_case("cfg2_mixed", "cfg2", "mixed")
    full_case("cfg3_first2_borderline", "cfg3", "borderline", with_fine=False, n=2)
    full_case("cfg5_borderline", "cfg5", "borderline", with_fine=False)
    # round 4: padding masks in the context layers, the L = 9600 cost volume (cfg#5 above now with its fine stage)
    masked_transformer_case()
    full_case("l9600_peaky", "l9600", "peaky")
    full_case("l9600_borderline", "l9600", "borderline", with_fine=False)
    # round 5: the a8 chain (net.py:66-83) at the size the bench times it at
    net_tail_case("net_tail_cfg2", NET_TAIL_CFG2)
    full_attention_case()
    masked_coarse_transformer_case()
    # round 6: the metric's own fine configuration (W = 5) and cfg#3 at its size (all 64 samples) through the reference
    w5_case("cfg2_peaky_w5", "cfg2", "peaky")
    w5_case("cfg2_borderline_w5", "cfg2", "borderline")
    batch_summary_case("cfg3_all64_peaky", "cfg3", "peaky")
    batch_summary_case("cfg3_all64_borderline", "cfg3", "borderline")
