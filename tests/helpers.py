"""Shared helpers for the parity tests (CPU side; no GPU, no reference access)."""
import os

import numpy as np

from featurematching_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLIPS = []      # (test id, guard-band flips, reference matches, max |conf - reference|) of every match-set comparison


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def load_kats(name="kats"):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cases = {}
    for k in z.files:
        case, key = k.split("/", 1)
        cases.setdefault(case, {})[key] = z[k]
    return cases


def case_inputs(meta, dist, with_fine=True, ww=49):
    """Regenerate the seeded inputs of a full_case fixture from its meta record."""
    n, h, w, c, cf, seed = [int(v) for v in meta]
    cfg = dict(n=n, h=h, w=w, c=c, cf=cf, seed=seed)
    sh = synth.config_shapes(cfg)
    f0, f1 = synth.coarse_descriptors(seed, n, sh['l'], c, dist)
    out = dict(cfg=cfg, sh=sh, f0=f0, f1=f1, hw_i=(h, w), hw_c=(sh['hc'], sh['wc']), hw_f=(sh['hf'], sh['wf']))
    if with_fine:
        out['ff0'], out['ff1'] = synth.fine_maps(seed, n, cf, sh['hf'], sh['wf'])
        out['mix'] = synth.mix_weights(seed, ww)
    return out


def compare_match_sets(got, ref, conf_tol=1e-5):
    """got/ref: dicts with b_ids,i_ids,j_ids,mconf (numpy).  Returns (only_got, only_ref, max_conf_err)
    where the first two are lists of (b,i,j,conf) present on one side only."""
    gk = {(int(b), int(i), int(j)): float(c) for b, i, j, c in zip(got['b_ids'], got['i_ids'], got['j_ids'], got['mconf'])}
    rk = {(int(b), int(i), int(j)): float(c) for b, i, j, c in zip(ref['b_ids'], ref['i_ids'], ref['j_ids'], ref['mconf'])}
    only_g = [(k, v) for k, v in gk.items() if k not in rk]
    only_r = [(k, v) for k, v in rk.items() if k not in gk]
    err = max([abs(gk[k] - rk[k]) for k in gk if k in rk], default=0.0)
    return only_g, only_r, err


NET_TAIL = dict(seed=31, n=2, h=128, w=128, c=256, cf=64, nhead=8, layers_c=['self', 'cross'] * 4,
                layers_f=['self', 'cross'], gain=1.0, sigma=1.6)


NET_TAIL_CFG2 = dict(NET_TAIL, seed=37, n=1, h=480, w=640)     # the same chain at the bench's size: one 640x480 pair


def net_tail_inputs(meta=NET_TAIL):
    """Seeded inputs of the net_tail_small fixture (tests/golden/make_golden.py:net_tail_inputs restated: feature
    maps as a backbone hands them over plus all weights, from the portable hash RNG)."""
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


def epipolar_inputs(seed=51, n=3, m=240):
    """Seeded matches / poses / intrinsics of the epi_small fixture (tests/golden/make_golden.py:epipolar_inputs
    restated)."""
    u = lambda st, shape: synth.uniform(seed, st, int(np.prod(shape))).reshape(shape)
    b = np.sort((u(1, (m,)) * n).astype(np.int64))
    k0 = np.concatenate([u(2, (m, 2)) * [640, 480], u(3, (m, 1))], 1).astype(np.float32)
    k1 = (k0 + np.concatenate([synth.normal(seed, 4, (m, 2)) * 6.0, np.zeros((m, 1))], 1)).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    for i in range(n):
        w = synth.normal(seed + i, 5, (3,)) * 0.2
        th = float(np.linalg.norm(w)); kx = w / th
        Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
        T[i, :3, :3] = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).astype(np.float32)
        T[i, :3, 3] = synth.normal(seed + i, 6, (3,)).astype(np.float32)
    K = np.tile(np.array([[500.0, 0, 320], [0, 510, 240], [0, 0, 1]], np.float32), (n, 1, 1))
    K0 = K + (u(7, (n, 3, 3)) * [[20, 0, 10], [0, 20, 10], [0, 0, 0]]).astype(np.float32)
    K1 = K + (u(8, (n, 3, 3)) * [[20, 0, 10], [0, 20, 10], [0, 0, 0]]).astype(np.float32)
    return dict(m_bids=b, mkpts0_f=k0, mkpts1_f=k1, T_0to1=T, K0=K0.astype(np.float32), K1=K1.astype(np.float32))
