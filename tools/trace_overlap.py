#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 kernel trace (…_kernel_trace.csv): how much of the wall time the
correlation sweeps run, how many kernels overlap, idle share.  Usage: trace_overlap.py FILE [skip_fraction]"""
import collections
import csv
import sys


def short(n):
    return n.replace('void fm::', '').replace('fm::', '').split('(')[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = 0.3
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
    t0, t1 = ev[int(len(ev) * skip)][0], ev[-40][1]
    pts = []
    for s, e, n in ev:
        if s >= t0 and e <= t1:
            pts.append((s, 1, n))
            pts.append((e, -1, n))
    pts.sort()
    active, conc, tot = collections.Counter(), collections.Counter(), collections.Counter()
    last = pts[0][0]
    for t, d, n in pts:
        dt = t - last
        if dt > 0:
            k = sum(active.values())
            conc[k] += dt
            for name, v in active.items():
                if v:
                    tot[name] += dt
            sweeps = sum(v for kk, v in active.items() if kk.startswith('k_corr<') and not kk.endswith('2>'))
            tot['[any sweep]'] += dt if sweeps else 0
            tot['[two sweeps]'] += dt if sweeps >= 2 else 0
        active[n] += d
        last = t
    w = t1 - t0
    print(f"window {w / 1e3:.0f} us")
    for k, v in sorted(conc.items()):
        print(f"  {k} kernels running: {v / w:6.3f}")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} active {v / w:6.3f} of the time")


if __name__ == "__main__":
    main()
