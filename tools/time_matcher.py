#!/usr/bin/env python3
"""Stage times of Matcher.forward_features (network/net.py:66-83) at 640x480 on one GPU: coarse context layers,
coarse matching, crop + context merge, fine context layers, fine matching.  Event-timed, eager, one pair."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import synth  # noqa: E402
from featurematching_amd.matcher import Matcher  # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--hc", type=int, default=60)
    ap.add_argument("--wc", type=int, default=80)
    ap.add_argument("--hip-only", action="store_true", help="skip the PyTorch-module comparison (counter runs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Matcher().to(dev).eval()
    n, hc, wc = a.n, a.hc, a.wc
    l = hc * wc
    f0, f1 = synth.coarse_descriptors(1, n, l, 256, "peaky")
    fc0 = torch.as_tensor(f0, device=dev).view(n, hc, wc, 256).permute(0, 3, 1, 2).contiguous()
    fc1 = torch.as_tensor(f1, device=dev).view(n, hc, wc, 256).permute(0, 3, 1, 2).contiguous()
    g = torch.Generator(device=dev).manual_seed(3)
    ff0 = torch.randn(n, 64, hc * 4, wc * 4, device=dev, generator=g)
    ff1 = torch.randn(n, 64, hc * 4, wc * 4, device=dev, generator=g)
    base = {'bs': n, 'hw0_i': (hc * 8, wc * 8), 'hw1_i': (hc * 8, wc * 8), 'hw0_c': (hc, wc), 'hw1_c': (hc, wc),
            'hw0_f': (hc * 4, wc * 4), 'hw1_f': (hc * 4, wc * 4)}
    with torch.no_grad():
        t_all, data = timed(lambda: m.forward_features(fc0, fc1, ff0, ff1, dict(base)))
        x0 = fc0.flatten(2).transpose(1, 2).contiguous()
        x1 = fc1.flatten(2).transpose(1, 2).contiguous()
        t_ctf, (c0, c1) = timed(lambda: m.coarse(x0, x1))
        # the matching stages on the descriptors themselves (the seeded-random context layers flatten them)
        d = dict(base)
        t_cm, _ = timed(lambda: m.coarse_matching(x0, x1, d))
        t_fp, (w0, w1) = timed(lambda: m.fine_preprocess(ff0, ff1, x0, x1, d))
        t_ftf, (v0, v1) = timed(lambda: m.fine(w0, w1))
        t_fm, _ = timed(lambda: m.fine_matching(v0, v1, d))
        t_ftf_t = float("nan")
        if not a.hip_only:
            m.fine.use_hip = False
            t_ftf_t, _ = timed(lambda: m.fine(w0, w1))
            m.fine.use_hip = True
    mm = int(d['b_ids'].numel())
    print(f"forward_features N={n} {hc * 8}x{wc * 8}: {t_all:.3f} ms per call (matches through the random context layers: "
          f"{int(data['b_ids'].numel())})")
    print(f"  coarse context layers (8 x [N,{l},256]) {t_ctf:8.3f} ms")
    print(f"  coarse matching (M={mm})               {t_cm:8.3f} ms")
    print(f"  crop + context merge                   {t_fp:8.3f} ms")
    print(f"  fine context layers, HIP               {t_ftf:8.3f} ms   (torch ops: {t_ftf_t:.3f} ms)")
    print(f"  fine matching                          {t_fm:8.3f} ms")


if __name__ == "__main__":
    main()
