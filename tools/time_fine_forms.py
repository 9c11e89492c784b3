#!/usr/bin/env python3
"""A/B in one process: the fine stage from NCHW maps by the copy form (fm_fine_match_maps: channels-last copy of image 1 +
k_fine_maps) and by the strip form (fm_fine_match_maps_cells: three strip passes, no copy), for several pairs per launch.

    python tools/time_fine_forms.py [--batches 1 4 8 16 64] [--window 5]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4, 8, 16, 64])
    ap.add_argument("--window", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(dev))          # (hipGraph capture needs a non-default stream)
    for n in a.batches:
        wl = dict(bench.WORKLOADS["cfg2"], n=n)
        p = bench.Pair(wl, 4242, a.window, dev, "peaky", device_data=True)
        buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, cell_maps=True)
        m = buf.read_count()
        scratch = torch.empty(p.ff1.numel() * 4 + 512, dtype=torch.uint8, device=dev)
        copy = lambda: ops.fine_match_maps(p.ff0, p.ff1, buf.b_ids, buf.i_ids, buf.j_ids, a.window, 4, p.hw_c[1], p.hw_c[1],
                                           p.mix0, p.mix1, buf.mkpts0_c, buf.mkpts1_c, 2.0, count=buf.count, scratch=scratch)
        strip = lambda: ops.fine_match_maps_cells(p.ff0, p.ff1, buf.b_ids, buf.i_ids, buf.j_ids, a.window, p.hw_c, p.hw_c,
                                                  p.mix0, p.mix1, buf.mkpts0_c, buf.mkpts1_c, 2.0, buf.cell_maps(),
                                                  count=buf.count, scratch=scratch)
        c0, c1 = copy()
        c0, c1 = c0.clone(), c1.clone()
        s0, s1 = strip()
        torch.cuda.synchronize()
        same = torch.equal(c0[:m], s0[:m]) and torch.equal(c1[:m], s1[:m])
        ts = {}
        for rnd in range(3):
            for name, fn in (("copy", copy), ("strip", strip)):
                ts.setdefault(name, []).append(bench._events(fn, group=3, iters=8 if n >= 16 else 30))
        print(f"pairs/launch {n:3d}  M {m:7d}  copy form {min(ts['copy']) * 1e3:9.1f} us   strip form {min(ts['strip']) * 1e3:9.1f} us   "
              f"identical {same}", flush=True)


if __name__ == "__main__":
    main()
