#!/usr/bin/env python3
"""In-kernel shader clock of the correlation sweeps (diagnostic build: tools/build_variant.sh diagclock
-DFM_DIAG_CLOCK).  Prints, per sweep, the cycles and the 100 MHz ticks wave 0 of the last panel's
workgroups spent inside the kernel body -> clock = cycles / ticks * 100 MHz.

    python tools/diag_clock.py [--dist peaky] build/variants/libfmatch_diagclock.so
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
    for name in ("fm_debug_launch_corr",):
        res, args = _lib.ALL_SIGNATURES[name]
        getattr(v, name).restype, getattr(v, name).argtypes = res, args
    buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)     # full-size workspace
    torch.cuda.synchronize()
    ws = buf.workspace
    off = (-ws.data_ptr()) % 256
    ptr = C.c_void_p(ws.data_ptr() + off)
    lay = layout(p.n, p.l, p.l, p.c, slots)
    names = ["cycles", "ticks100MHz", "units", "chain_cyc", "-", "barrier_cyc", "prologue_cyc", "tail_cyc"]
    for mode in (1,):
        for rep in range(20):          # warm: the clock ramps with load
            lib.fm_debug_reset_counters(ptr, p.n, p.l, p.l, p.c, slots, st)
            # (the screening kernel flags the samples the dense kernel redoes)
            lib.fm_debug_launch_screen(ptr, C.c_void_p(p.f0.data_ptr()), C.c_void_p(p.f1.data_ptr()), p.n, p.l, p.l, p.c,
                                           slots, 0.1, 0.2, st)
            v.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, mode, st)
        torch.cuda.synchronize()
        o = off + lay["cand_x"] + p.l * slots * 4             # candidate slots of the first padded row
        d = ws[o: o + 512 * 4].view(torch.float32).cpu().numpy().reshape(64, 8)     # [panel*8 + wave, stamp]
        print(f"mode {mode}: {d.shape[0]} waves of split 0 ({lay['splits']} splits, "
              f"{-(-lay['tiles'] // lay['splits'])} tiles each); clock {np.median(d[:, 0]) / np.median(d[:, 1]) * 0.1:.2f} GHz")
        for k, nme in enumerate(names):
            col = d[:, k]
            print(f"   {nme:14s} median {np.median(col):9.0f}  mean {col.mean():9.0f}  min {col.min():9.0f}  max {col.max():9.0f}")
        early, late = d[np.arange(d.shape[0]) % 8 < 4], d[np.arange(d.shape[0]) % 8 >= 4]
        print(f"   per computed unit: mfma {d[:, 3].sum() / max(d[:, 2].sum(), 1):.0f} cyc, epilogue {d[:, 4].sum() / max(d[:, 2].sum(), 1):.0f} cyc"
              f"   (early waves epi {early[:, 4].sum() / max(early[:, 2].sum(), 1):.0f}, late {late[:, 4].sum() / max(late[:, 2].sum(), 1):.0f})")


if __name__ == "__main__":
    main()
