#!/usr/bin/env python3
"""Times the correlation sweeps of experimental builds (tools/build_variant.sh) on a workspace prepared
by the shipped library, so that timing-only ablations see the real stabilisers and block map.

    python tools/time_corr_variants.py [--workload cfg2] [--dist peaky] build/variants/libfmatch_X.so ...
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402


def timed(fn, iters, pre=None):
    ts = []
    for _ in range(iters):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--dist", default="peaky")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    p = bench.Pair(bench.WORKLOADS[a.workload], 1017, 5, dev, a.dist)
    lib = _lib.load()
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"workload {a.workload} dist {a.dist}: N={p.n} L={p.l} C={p.c}")
    for path in ["<shipped>"] + a.libs:
        buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, dense=True)     # full-size workspace   # fresh, correct workspace
        torch.cuda.synchronize()
        ptr = C.c_void_p(buf.workspace.data_ptr() + ((-buf.workspace.data_ptr()) % 256))
        v = lib
        if path != "<shipped>":
            v = C.CDLL(os.path.abspath(path))
            for name in ("fm_debug_launch_corr", "fm_debug_reset_counters"):
                res, args = _lib.ALL_SIGNATURES[name]
                getattr(v, name).restype, getattr(v, name).argtypes = res, args
        # sum pass first (needs the real block map), max pass last (an ablated one may leave the map stale)
        s_med, s_min = timed(lambda: v.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 1, st), a.iters,
                             lambda: lib.fm_debug_reset_counters(ptr, p.n, p.l, p.l, p.c, slots, st))
        m_med, m_min = timed(lambda: v.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 0, st), a.iters)
        torch.cuda.synchronize()
        print(f"{os.path.basename(path):40s} max pass {m_med:8.1f} us (min {m_min:8.1f})   sum pass {s_med:8.1f} us (min {s_min:8.1f})")


if __name__ == "__main__":
    main()
