#!/usr/bin/env python3
"""In-kernel shader-clock stamps of the int8 max pass (diagnostic build: tools/build_variant.sh diagclock
-DFM_DIAG_CLOCK): per wave the cycles of the whole kernel, the prologue (A fragments + first tiles), the MFMA
chains, the epilogues, the tile barriers and the LDS-DMA issue.

    python tools/diag_max.py [--dist peaky] [--workload cfg2] build/variants/libfmatch_diagclock.so
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
    ap.add_argument("--reps", type=int, default=20, help="back-to-back launches before the stamps are read")
    ap.add_argument("lib")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    p = bench.Pair(bench.WORKLOADS[a.workload], 1017, 5, dev, a.dist)
    lib = _lib.load()
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    v = C.CDLL(os.path.abspath(a.lib))
    res, args = _lib.ALL_SIGNATURES["fm_debug_launch_corr"]
    v.fm_debug_launch_corr.restype, v.fm_debug_launch_corr.argtypes = res, args
    buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)     # full-size workspace
    torch.cuda.synchronize()
    ws = buf.workspace
    off = (-ws.data_ptr()) % 256
    ptr = C.c_void_p(ws.data_ptr() + off)
    lay = layout(p.n, p.l, p.l, p.c, slots)
    for rep in range(a.reps):
        v.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 0, st)
    torch.cuda.synchronize()
    tiles = lay["Sp"] // 64                                   # the max pass' own grid (api.hip: kMaxPassTarget)
    wg = p.n * lay["panels"]                                  # (api.hip: choose_splits with kMaxPassTarget / kMaxPassSlots)
    s0 = min(max((5120 if wg >= 512 else 256) // wg, 1), max(tiles // 2, 1), 32)
    tps = -(-tiles // s0)
    msplits = -(-tiles // tps)
    nwg = p.n * lay["panels"] * msplits
    o = off + lay["rowB"]
    d = ws[o: o + nwg * 4 * 8 * 4].view(torch.float32).cpu().numpy().reshape(nwg * 4, 8)
    names = ["total", "prologue", "mfma", "epilogue", "barrier", "realtime", "start", "tail"]
    print(f"{nwg} workgroups x 4 waves ({msplits} splits x {tps} tiles)")
    units = 2 * tps
    st0 = d[:, 6].copy()
    d[:, 6] = ((st0 - st0.min()) % (1 << 24))            # start of the wave after the first wave's, 10 ns ticks
    for k, nme in enumerate(names):
        col = d[:, k]
        print(f"   {nme:9s} median {np.median(col):8.0f}  mean {col.mean():8.0f}  min {col.min():8.0f}  max {col.max():8.0f}")
    print(f"   in-kernel clock (total cycles / 10 ns ticks): median {np.median(d[:, 0] / np.maximum(d[:, 5], 1)) * 100:.0f} MHz")
    order = np.argsort(-d[:, 5])[:16]
    print("   slowest waves (workgroup.wave: start, realtime | total prologue mfma epilogue barrier tail):")
    for q in order:
        print(f"     {q // 4:4d}.{q % 4}: {d[q, 6]:5.0f} {d[q, 5]:6.0f} | " + " ".join(f"{d[q, k]:7.0f}" for k in (0, 1, 2, 3, 4, 7)))
    print(f"   last wave ends {np.max(d[:, 6] + d[:, 5]) / 100:.2f} us after the first wave starts")
    print(f"   per unit: mfma {d[:, 2].sum() / (units * len(d)):.0f} cyc, epilogue {d[:, 3].sum() / (units * len(d)):.0f} cyc")


if __name__ == "__main__":
    main()
