#!/usr/bin/env python3
"""Constant-clock stamps (10 ns ticks) of k_select's phases per workgroup (diagnostic build: tools/build_variant.sh
diagclock -DFM_DIAG_CLOCK): ticket, loads + conf, publish, look-back, emit.

    python tools/diag_select.py build/variants/libfmatch_diagclock.so
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402
from tools.gpu_bringup import layout  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="peaky")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("lib")
    a = ap.parse_args()
    _lib.load(os.path.abspath(a.lib))          # the diagnostic build IS the library of this process
    dev = torch.device("cuda:0")
    p = bench.Pair(bench.WORKLOADS[a.workload], 1017, 5, dev, a.dist)
    lib = _lib.load()
    slots = lib.fm_default_cand_slots(0.2)
    for _ in range(5):
        buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)
    torch.cuda.synchronize()
    ws = buf.workspace
    off = (-ws.data_ptr()) % 256
    lay = layout(p.n, p.l, p.l, p.c, slots)
    nblk = (p.n * lay["Lp"] * slots + 255) // 256
    o = off + lay["rowB"]
    d = ws[o: o + nblk * 8 * 4].view(torch.float32).cpu().numpy().reshape(nblk, 8)[:, :6]
    t0 = d[:, 0].min()
    rel = ((d - t0) % (1 << 24)) * 0.01          # microseconds since the first workgroup started
    names = ["start", "ticket", "loads+conf", "published", "look-back", "end"]
    print(f"{nblk} workgroups; times in us since the first workgroup's start")
    for k, nme in enumerate(names):
        col = rel[:, k]
        print(f"   {nme:11s} median {np.median(col):6.2f}  min {col.min():6.2f}  max {col.max():6.2f}")
    seg = np.diff(rel, axis=1)
    for k, nme in enumerate(["ticket", "loads+conf", "publish", "look-back", "emit"]):
        print(f"   d[{nme:10s}] median {np.median(seg[:, k]):6.2f}  max {seg[:, k].max():6.2f}")
    # k_prep_split of the same call: stamps in the dense kernel's column partials
    nb = p.n * (lay["Lp"] + lay["Sp"]) // 32
    o = off + lay["colB"]
    d = ws[o: o + nb * 8 * 4].view(torch.float32).cpu().numpy().reshape(nb, 8)[:, :5]
    rel = ((d - d[:, 0].min()) % (1 << 24)) * 0.01
    print(f"k_prep_split: {nb} workgroups")
    for k, nme in enumerate(["start", "loads issued", "loads landed", "quantised", "end"]):
        col = rel[:, k]
        print(f"   {nme:13s} median {np.median(col):6.2f}  min {col.min():6.2f}  max {col.max():6.2f}")


if __name__ == "__main__":
    main()
