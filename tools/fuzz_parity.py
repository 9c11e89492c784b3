#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite: minutes of oracle time): random cell grids, channel counts, batch
sizes, thresholds, border widths and data kinds through ops.coarse_match + fine_match_maps against the CPU oracle.

    python tools/fuzz_parity.py [--cases 60] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from featurematching_amd import ops, synth  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (checker)
from helpers import compare_match_sets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="run these case numbers only")
    ap.add_argument("--conf", action="store_true", help="also request data['conf_matrix'] and compare the whole matrix (T = 0.1 cases)")
    ap.add_argument("--ties", action="store_true", help="duplicate a few descriptors in either image: exactly tied conf entries "
                                                       "(coarse_matching_new.py:105-106 keeps all of them)")
    ap.add_argument("--half", action="store_true", help="hand the descriptors over as float16 / bfloat16 (oracle on the up-cast values)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    bad = 0
    for case in range(a.cases):
        h0, w0 = int(rng.integers(5, 41)), int(rng.integers(5, 41))
        same = rng.random() < 0.6
        h1, w1 = (h0, w0) if same else (int(rng.integers(5, 41)), int(rng.integers(5, 41)))
        c = int(rng.choice([32, 64, 96, 128, 200, 256]))
        n = int(rng.integers(1, 5))
        thr = float(rng.choice([0.2, 0.2, 0.1, 0.35, 0.05]))
        border = int(rng.integers(0, 3))
        temp = float(rng.choice([0.1, 0.1, 0.2, 0.05]))
        dist = str(rng.choice(["peaky", "borderline", "mixed"]))
        w = int(rng.choice([5, 7]))
        if a.only is not None and case not in a.only:
            continue
        l, s_ = h0 * w0, h1 * w1
        lm = max(l, s_)
        f0, f1 = synth.coarse_descriptors(1000 + case, n, lm, c, dist)
        f0, f1 = np.ascontiguousarray(f0[:, :l]), np.ascontiguousarray(f1[:, :s_])
        if l != s_:          # rectangular: the partners of the rows beyond min(l, s) are missing
            pass
        if a.ties:
            for img, length in ((f1, s_), (f0, l)):
                for _ in range(int(rng.integers(0, 4))):
                    b_, src, dst = int(rng.integers(0, n)), int(rng.integers(0, length)), int(rng.integers(0, length))
                    img[b_, dst] = img[b_, src]
        hw_i = (8 * h0, 8 * w0)
        hdt = None
        if a.half:
            hdt = torch.float16 if case % 2 else torch.bfloat16
            f0 = torch.as_tensor(f0).to(hdt).float().numpy()      # the values the half-precision tensors hold
            f1 = torch.as_tensor(f1).to(hdt).float().numpy()
        ref = orc.coarse_match(f0, f1, hw_i, (h0, w0), (h1, w1), thr=thr, border_rm=border, temperature=temp)
        t0, t1 = torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev)
        if hdt is not None:
            t0, t1 = t0.to(hdt), t1.to(hdt)
        want_conf = a.conf and temp >= 0.1
        out = ops.coarse_match(t0, t1, (h0, w0), (h1, w1), hw_i[0] / h0, thr, border, temp, conf_matrix=want_conf)
        got = {k: v.cpu().numpy() for k, v in out.items() if not k.startswith('_') and k != 'conf_matrix'}
        r = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in ref.items()}
        og, orf, err = compare_match_sets(got, r)
        flips = [(k, v) for k, v in og + orf if abs(v - thr) > 2e-5]
        msg = f"case {case:3d} n={n} {h0}x{w0}/{h1}x{w1} C={c} {dist:10s} thr={thr} b={border} T={temp} W={w}: M={len(r['i_ids'])} conf err {err:.1e}"
        ok = not flips and err <= 1e-5
        if want_conf:
            cerr = float((out['conf_matrix'].cpu() - orc.conf_matrix(torch.as_tensor(f0), torch.as_tensor(f1), temp)).abs().max())
            msg += f" conf_matrix err {cerr:.1e}"
            ok = ok and cerr <= 1e-5
        if not flips and err > 1e-5:
            # beyond the bar against the float32 oracle: where does float64 put the entries?  (the reference's own float32
            # sums are up to ~1e-5 from float64 at low temperatures / large S: BASELINE.md section 4's exception)
            a0 = torch.as_tensor(f0, dtype=torch.float64); a1 = torch.as_tensor(f1, dtype=torch.float64)
            sim = torch.einsum("nlc,nsc->nls", a0 / c ** .5, a1 / c ** .5) / temp
            c64 = (torch.softmax(sim, 1) * torch.softmax(sim, 2))
            gk = {(int(b_), int(i_), int(j_)): float(v) for b_, i_, j_, v in zip(got['b_ids'], got['i_ids'], got['j_ids'], got['mconf'])}
            rk = {(int(b_), int(i_), int(j_)): float(v) for b_, i_, j_, v in zip(r['b_ids'], r['i_ids'], r['j_ids'], r['mconf'])}
            e_hip = max(abs(v - float(c64[k])) for k, v in gk.items())
            e_ref = max(abs(v - float(c64[k])) for k, v in rk.items())
            msg += f" | vs float64: HIP {e_hip:.1e}, float32 oracle {e_ref:.1e}"
            ok = e_hip <= 5e-6
        if ok and len(got['i_ids']) and not og and not orf:
            ff0, _ = synth.fine_maps(2000 + case, n, 64, 4 * h0, 4 * w0)
            _, ff1 = synth.fine_maps(2000 + case, n, 64, 4 * h1, 4 * w1)
            mix = synth.mix_weights(case, w * w)
            mix0 = torch.as_tensor(np.concatenate([mix[0], [mix[1]]]).astype(np.float32), device=dev)
            mix1 = torch.as_tensor(np.concatenate([mix[2], [mix[3]]]).astype(np.float32), device=dev)
            k0, k1 = ops.fine_match_maps(torch.as_tensor(ff0, device=dev), torch.as_tensor(ff1, device=dev), out['b_ids'],
                                         out['i_ids'], out['j_ids'], w, 4, w0, w1, mix0, mix1, out['mkpts0_c'], out['mkpts1_c'], 2.0)
            w0t = orc.crop_windows(ff0, ref['b_ids'], ref['i_ids'], w, 4, w0)
            w1t = orc.crop_windows(ff1, ref['b_ids'], ref['j_ids'], w, 4, w1)
            r0, r1 = orc.fine_match(w0t, w1t, mix[0], mix[1], mix[2], mix[3], ref['mkpts0_c'], ref['mkpts1_c'], 2.0)
            fe = max(float((k0.cpu() - r0).abs().max()), float((k1.cpu() - r1).abs().max()))
            msg += f" fine err {fe:.1e}"
            ok = ok and fe <= 1e-3
            # the other routes to the same numbers: list-ordered crops + fm_fine_match, the cell-ordered pair crop, channels-last maps
            tf0, tf1 = torch.as_tensor(ff0, device=dev), torch.as_tensor(ff1, device=dev)
            win0 = ops.gather_windows(tf0, out['b_ids'], out['i_ids'], w, 4, w0)
            win1 = ops.gather_windows(tf1, out['b_ids'], out['j_ids'], w, 4, w1)
            q0, q1 = ops.fine_match(win0, win1, mix0, mix1, out['mkpts0_c'], out['mkpts1_c'], 2.0)
            p0, p1 = ops.gather_windows_pair(tf0, tf1, out['b_ids'], out['i_ids'], out['j_ids'], w, 4, (h0, w0), (h1, w1),
                                             out['_coarse_buffers'].cell_maps())
            c0, c1 = ops.fine_match_maps(tf0.contiguous(memory_format=torch.channels_last), tf1.contiguous(memory_format=torch.channels_last),
                                         out['b_ids'], out['i_ids'], out['j_ids'], w, 4, w0, w1, mix0, mix1, out['mkpts0_c'], out['mkpts1_c'], 2.0)
            same = (torch.equal(q0, k0) and torch.equal(q1, k1) and torch.equal(p0, win0) and torch.equal(p1, win1)
                    and torch.equal(c0, k0) and torch.equal(c1, k1)
                    and torch.equal(win0.cpu(), w0t) and torch.equal(win1.cpu(), w1t))
            msg += " routes identical" if same else " ROUTES DIFFER"
            ok = ok and same
        print(("ok   " if ok else "FAIL ") + msg + (f" flips {flips[:3]}" if flips else ""), flush=True)
        bad += 0 if ok else 1
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
