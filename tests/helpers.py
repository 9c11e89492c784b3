"""Shared helpers for the parity tests (CPU side; no GPU, no reference access)."""
import os

import numpy as np

from featurematching_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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
