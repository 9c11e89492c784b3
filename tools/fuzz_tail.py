#!/usr/bin/env python3
"""Randomised sweep of the whole tail of net.forward (network/net.py:66-83) - Matcher.forward_features: HIP coarse context
layers -> coarse matching -> crop + context merge -> fine context layers -> fine matching - against the oracle's restatement
(oracle.net_tail, pinned by the net_tail_* reference fixtures) on random image sizes, batch sizes, seeds and noise levels.

    python tools/fuzz_tail.py [--cases 12] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from featurematching_amd.matcher import Matcher  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (checker)
from helpers import NET_TAIL, compare_match_sets, net_tail_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    DEV = "cuda:0"
    torch.set_num_threads(16)
    bad = 0
    for case in range(a.cases):
        meta = dict(NET_TAIL, seed=int(rng.integers(100, 10000)), n=int(rng.integers(1, 4)), h=8 * int(rng.integers(8, 33)),
                    w=8 * int(rng.integers(8, 33)), sigma=float(rng.choice([1.2, 1.6, 2.0])))
        inp = net_tail_inputs(meta)
        ref = orc.net_tail(inp['feat_c0'], inp['feat_c1'], inp['feat_f0'], inp['feat_f1'], inp['hw_i'], inp['w_coarse'], inp['w_fine'],
                           inp['w_prep'], inp['mix'], meta['layers_c'], meta['layers_f'], nhead=meta['nhead'])
        m = Matcher().to(DEV).eval()
        t = lambda d: {k: torch.as_tensor(v) for k, v in d.items()}
        m.coarse.load_state_dict(t(inp['w_coarse'])); m.fine.load_state_dict(t(inp['w_fine'])); m.fine_preprocess.load_state_dict(t(inp['w_prep']))
        w0, b0, w1, b1 = inp['mix']
        with torch.no_grad():
            m.fine_matching.mix_feat_0.weight.copy_(torch.as_tensor(w0).view(1, -1)); m.fine_matching.mix_feat_0.bias.fill_(float(b0))
            m.fine_matching.mix_feat_1.weight.copy_(torch.as_tensor(w1).view(1, -1)); m.fine_matching.mix_feat_1.bias.fill_(float(b1))
        dev = lambda x: torch.as_tensor(x, device=DEV)
        data = {'bs': meta['n'], 'hw0_i': inp['hw_i'], 'hw1_i': inp['hw_i']}
        m.forward_features(dev(inp['feat_c0']), dev(inp['feat_c1']), dev(inp['feat_f0']), dev(inp['feat_f1']), data)
        got = {k: data[k].cpu().numpy() for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_c', 'mkpts1_c')}
        r = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in ref.items()}
        og, orf, err = compare_match_sets(got, r)
        flips = [(k, v) for k, v in og + orf if abs(v - 0.2) > 4e-5]
        gk = {(int(b), int(i), int(j)): n for n, (b, i, j) in enumerate(zip(got['b_ids'], got['i_ids'], got['j_ids']))}
        rk = {(int(b), int(i), int(j)): n for n, (b, i, j) in enumerate(zip(r['b_ids'], r['i_ids'], r['j_ids']))}
        common = [k for k in gk if k in rk]
        fe = 0.0
        if common:
            gi, ri = np.array([gk[k] for k in common]), np.array([rk[k] for k in common])
            fe = max(float(np.abs(data['mkpts0_f'].cpu().numpy()[gi, :2] - r['mkpts0_f'][ri, :2]).max()),
                     float(np.abs(data['mkpts1_f'].cpu().numpy()[gi, :2] - r['mkpts1_f'][ri, :2]).max()))
        ok = not flips and err <= 4e-5 and fe <= 5e-4
        bad += 0 if ok else 1
        print(("ok   " if ok else "FAIL ") + f"case {case:2d} n={meta['n']} {meta['h']}x{meta['w']} sigma={meta['sigma']} seed={meta['seed']}: M={len(r['i_ids'])} "
              f"guard-band flips {len(og) + len(orf)} conf err {err:.1e} fine err {fe:.1e} px range fallbacks {int(m.fine.range_fallbacks)}"
              + (f" OUTSIDE the band: {flips[:3]}" if flips else ""), flush=True)
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
