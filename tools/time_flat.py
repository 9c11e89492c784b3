#!/usr/bin/env python3
"""FM_MODE_FLAT against FM_MODE_DENSE alone on flat data: kernel times of one 640x480 pair (events, one stream) and the
4-stream rate of the whole step, each measured twice in alternating order (same process, same device)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["cfg2"]
    for dist in ("borderline", "mixed"):
        for flat in (True, False):
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                p = bench.Pair(wl, 7777, 5, dev, dist)
                p.flat = flat
                p.step()
                torch.cuda.synchronize()
                p.last[0].read_count()
                t = bench.time_kernels(p)
            print(dist, "flat" if flat else "dense", {k: round(v * 1e3, 2) for k, v in t.items()}, flush=True)
            del p
        for rep in range(2):
            for flat in (True, False):
                r, ver, _ = bench.stream_rate(wl, 5, dev, dist, 1, 4, steps=400, nsets=8, flat_hint=flat)
                r1, _, _ = bench.stream_rate(wl, 5, dev, dist, 1, 1, steps=200, nsets=4, flat_hint=flat, check=False)
                print(f"  {dist} rep {rep} {'flat ' if flat else 'dense'}: 4 streams {r:9.1f} pairs/s, 1 stream {r1:9.1f}  verified {ver['ok'] if ver else None}", flush=True)


def slots_experiment():
    """'mixed' data under the flat hint: more candidate slots instead of the exact screening pass"""
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["cfg2"]
    for slots, exact, es in ((8, True, False), (8, False, True), (8, True, True), (16, False, True), (8, True, False), (8, False, True)):
        try:
            r, ver, _ = bench.stream_rate(wl, 5, dev, "mixed", 1, 4, steps=400, nsets=8, slots=slots, exact=exact, exact_step=es)
            r1, _, _ = bench.stream_rate(wl, 5, dev, "mixed", 1, 1, steps=200, nsets=4, slots=slots, exact=exact, exact_step=es, check=False)
            print(f"  mixed slots {slots} exact screening {exact} exact step {es}: 4 streams {r:9.1f}, 1 stream {r1:9.1f}, verified {ver['ok'] if ver else None}", flush=True)
        except Exception as e:
            print(f"  mixed slots {slots} exact {exact}: {e!r}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "slots":
        slots_experiment()
        sys.exit(0)
    main()
