#!/usr/bin/env python3
"""A/B of the max pass (k_max_i8) of two builds of the library in ONE process, interleaved rounds (devices and runs
differ by ~10 %: never compare numbers of different gpurun calls):

    python tools/ab_max.py build/variants/libfmatch_A.so build/variants/libfmatch_B.so [--workloads cfg2 cfg3 cfg5]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workloads", nargs="+", default=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--mode", type=int, default=0, help="fm_debug_launch_corr mode (0 = max pass)")
    ap.add_argument("--kernel", default="corr", choices=["corr", "sparse", "prep"],
                    help="corr: fm_debug_launch_corr(--mode); sparse: counter reset + fm_debug_launch_screen")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--env", default="", help="NAME=v1,v2,..: every library is timed once per value (FM_TUNE_ENV builds read "
                                              "their tuning variables at every call)")
    ap.add_argument("--batch", type=int, nargs="*", default=[], help="pairs per launch to time instead of the workloads' own (cfg2-sized pairs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res, args = _lib.ALL_SIGNATURES["fm_debug_launch_corr"]
    vs, names, envs = [], [], []
    ename, evals = (a.env.split("=")[0], a.env.split("=")[1].split(",")) if a.env else (None, [None])
    for path in a.libs:
        v = C.CDLL(os.path.abspath(path))
        v.fm_debug_launch_corr.restype, v.fm_debug_launch_corr.argtypes = res, args
        for fn in ("fm_debug_launch_screen", "fm_debug_reset_counters", "fm_debug_launch_prep"):
            getattr(v, fn).restype, getattr(v, fn).argtypes = _lib.ALL_SIGNATURES[fn]
        for ev in evals:
            vs.append(v); envs.append(ev)
            names.append(os.path.basename(path) + (f" {ename}={ev}" if ev else ""))
    libc = C.CDLL(None)
    todo = [(wl, bench.WORKLOADS[wl]) for wl in a.workloads] if not a.batch else \
        [(f"cfg2x{n}", dict(bench.WORKLOADS["cfg2"], n=n)) for n in a.batch]
    for wl, wld in todo:
        p = bench.Pair(wld, 1017, 5, dev, "peaky", device_data=True)
        buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)
        torch.cuda.synchronize()
        ws = buf.workspace
        ptr = C.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)
        n = 200 if p.n * p.l <= 4 * 4800 else 30
        t = [[] for _ in vs]
        for rnd in range(a.rounds + 1):
            for k, v in enumerate(vs):
                if envs[k] is not None:
                    libc.setenv(ename.encode(), envs[k].encode(), 1)      # (os.environ alone may not reach getenv of the C side)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                f0p, f1p = C.c_void_p(p.f0.data_ptr()), C.c_void_p(p.f1.data_ptr())
                for _ in range(n):
                    if a.kernel == "prep":
                        v.fm_debug_launch_prep(ptr, f0p, f1p, p.n, p.l, p.l, p.c, slots, st)
                    elif a.kernel == "corr":
                        v.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, a.mode, st)
                    else:
                        v.fm_debug_reset_counters(ptr, p.n, p.l, p.l, p.c, slots, st)
                        v.fm_debug_launch_screen(ptr, f0p, f1p, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, st)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    t[k].append(e0.elapsed_time(e1) * 1e3 / n)
        for k, nm in enumerate(names):
            print(f"{wl} {nm:44s} median {np.median(t[k]):9.2f} us  min {min(t[k]):9.2f} us")


if __name__ == "__main__":
    main()
