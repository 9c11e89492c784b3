#!/usr/bin/env python3
"""In-kernel shader-clock stamps of the sparse sum kernel (diagnostic build: tools/build_variant.sh diagclock
-DFM_DIAG_CLOCK).  Per wave: cycles of (1) statistics round trip, (2) stabilisers + barriers + live mask,
(3) first B fragments landed, (4) sweep over the live units, (5) exact dot products, (6) stores + fold.

    python tools/diag_sparse.py [--dist peaky] [--workload cfg2] build/variants/libfmatch_diagclock.so
"""
import argparse
import ctypes as C
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
    dev = torch.device("cuda:0")
    p = bench.Pair(bench.WORKLOADS[a.workload], 1017, 5, dev, a.dist)
    lib = _lib.load()
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    v = C.CDLL(os.path.abspath(a.lib))
    for name in ("fm_debug_launch_screen", "fm_debug_reset_counters"):
        res, args = _lib.ALL_SIGNATURES[name]
        getattr(v, name).restype, getattr(v, name).argtypes = res, args
    buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)     # full-size workspace
    torch.cuda.synchronize()
    ws = buf.workspace
    off = (-ws.data_ptr()) % 256
    ptr = C.c_void_p(ws.data_ptr() + off)
    lay = layout(p.n, p.l, p.l, p.c, slots)
    for rep in range(20):
        v.fm_debug_reset_counters(ptr, p.n, p.l, p.l, p.c, slots, st)
        v.fm_debug_launch_screen(ptr, C.c_void_p(p.f0.data_ptr()), C.c_void_p(p.f1.data_ptr()), p.n, p.l, p.l, p.c,
                                     slots, 0.1, 0.2, st)
    torch.cuda.synchronize()
    nwg = p.n * lay["panels"] * lay["splits_s"]
    o = off + lay["rowB"]
    d = ws[o: o + nwg * 8 * 8 * 4].view(torch.float32).cpu().numpy().reshape(nwg * 8, 8)
    names = ["stats_rt", "stab+mask", "first_B", "sweep", "exact", "store+fold", "live_units", "entries"]
    print(f"{nwg} workgroups x 8 waves ({lay['splits_s']} splits x {lay['units_s']} units)")
    for k, nme in enumerate(names):
        col = d[:, k]
        print(f"   {nme:11s} median {np.median(col):8.0f}  mean {col.mean():8.0f}  min {col.min():8.0f}  max {col.max():8.0f}")
    tot = d[:, :6].sum(1)
    print(f"   total       median {np.median(tot):8.0f}  mean {tot.mean():8.0f}  max {tot.max():8.0f}")
    wg = tot.reshape(nwg, 8).max(1)
    print("   per-wave total percentiles 50/90/99/100:", [int(np.percentile(tot, q)) for q in (50, 90, 99, 100)])
    print("   per-workgroup (slowest wave) percentiles 50/90/99/100:", [int(np.percentile(wg, q)) for q in (50, 90, 99, 100)])
    worst = int(np.argmax(tot))
    print("   slowest wave:", [int(x) for x in d[worst]])
    live = d[:, 6] > 0
    print(f"   sweep per live unit {d[live, 3].sum() / d[live, 6].sum():.0f} cyc; exact per entry {d[live, 4].sum() / max(d[live, 7].sum(), 1):.0f} cyc")


if __name__ == "__main__":
    main()
