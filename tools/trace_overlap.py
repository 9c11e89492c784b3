#!/usr/bin/env python3
"""Concurrency of the kernels of a multi-stream bench run from a rocprofv3 --kernel-trace CSV:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --quick --skip-cpu --steps 600 --reps 2
    python tools/trace_overlap.py gpurun_out/trace

per kernel: launches, mean duration; per queue: mean gap between consecutive kernels; how many kernels run at once
(time-weighted) over the densest part of the trace."""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def main():
    root = sys.argv[1]
    f = max(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if not n.startswith("void fm::") and "fm::" not in n:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("fm::")[-1].split("(")[0],
                         r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    # the densest 10 ms of the trace (a bench run is mostly host-side pauses between its timed regions)
    starts = np.array([r[0] for r in rows])
    win = 10_000_000
    best, at = 0, 0
    for i in range(0, len(rows), 50):
        n = int(np.searchsorted(starts, starts[i] + win)) - i
        if n > best:
            best, at = n, i
    rows = [r for r in rows[at:at + best]]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"{len(rows)} launches over {(t1 - t0) / 1e3:.0f} us")
    dur = defaultdict(list)
    for s, e, n, q, st in rows:
        dur[n].append((e - s) / 1e3)
    tot = 0.0
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:28s} {len(v):6d} launches  mean {np.mean(v):7.2f} us  median {np.median(v):7.2f}  sum/wall {sum(v) * 1e3 / (t1 - t0):.3f}")
        tot += sum(v)
    print(f"  sum of durations / wall = {tot * 1e3 / (t1 - t0):.2f} kernels running on average")
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    lvl, last, hist = 0, ev[0][0], defaultdict(float)
    for t, d in ev:
        hist[lvl] += t - last
        last = t
        lvl += d
    wall = sum(hist.values())
    print("  kernels running at once: " + "  ".join(f"{k}: {v / wall:.3f}" for k, v in sorted(hist.items())))
    byq = defaultdict(list)
    for s, e, n, q, st in rows:
        byq[(q, st)].append((s, e, n))
    gaps = defaultdict(list)
    for q, v in byq.items():
        for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:]):
            gaps[f"{n0} -> {n1}"].append((s1 - e0) / 1e3)
    for q, v in sorted(byq.items()):
        span = v[-1][1] - v[0][0]
        print(f"  queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / span:.3f} of its span")
    print(f"  {len(byq)} queues/streams; gaps between consecutive kernels of a queue:")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
        print(f"    {k:60s} n {len(v):5d}  median {np.median(v):6.2f} us  mean {np.mean(v):6.2f}")


if __name__ == "__main__":
    main()
