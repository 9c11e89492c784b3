#!/usr/bin/env python3
"""How well does each kernel of the step overlap with copies of ITSELF on other streams?  S streams (own inputs and
workspace each) replay a graph of 8 launches of one kernel; a kernel that leaves room on the chip keeps its per-launch
time as S grows (job time per launch ~ 1/S), one that fills the chip does not.  Three normal-priority streams have a
hardware queue each (tools/probe_stream_queues.py), so S <= 3 measures the kernels, not the queues."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
slots = lib.fm_default_cand_slots(0.2)
wl = bench.WORKLOADS["cfg2"]
NS = 3
pairs = [bench.Pair(wl, 1000 + 17 * p, 5, dev, "peaky") for p in range(NS)]
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
bufs = []
for p, st in zip(pairs, streams):
    with torch.cuda.stream(st):
        bufs.append(p.step())
torch.cuda.synchronize()


def ws_ptr(buf):
    ws = buf[0].workspace
    return C.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)


def kernels(k):
    p, buf = pairs[k], bufs[k][0]
    sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f0, f1 = C.c_void_p(p.f0.data_ptr()), C.c_void_p(p.f1.data_ptr())
    return {
        "k_prep_split": lambda: lib.fm_debug_launch_prep(ws_ptr(bufs[k]), f0, f1, p.n, p.l, p.l, p.c, slots, sp()),
        "k_max_i8": lambda: lib.fm_debug_launch_corr(ws_ptr(bufs[k]), p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 0, sp()),
        "k_sum_sparse": lambda: (lib.fm_debug_reset_counters(ws_ptr(bufs[k]), p.n, p.l, p.l, p.c, slots, sp()),
                                 lib.fm_debug_launch_screen(ws_ptr(bufs[k]), f0, f1, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, sp())),
        "coarse stage (4 kernels)": lambda: ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, cell_maps=False),
        "transpose + k_fine_maps": lambda: p.fine_maps(buf),
    }


names = list(kernels(0).keys())
print(f"{'kernel':28s}" + "".join(f"  S={s}: us/launch (x)" for s in range(1, NS + 1)))
for name in names:
    graphs = []
    for k in range(NS):
        fn = kernels(k)[name]
        with torch.cuda.stream(streams[k]):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[k]):
            for _ in range(8):
                fn()
        graphs.append(g)
    res = []
    for S in range(1, NS + 1):
        def go():
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        R = 60
        t0 = time.perf_counter()
        for _ in range(R):
            go()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 1e6 / (R * 8 * S))
    print(f"{name:28s}" + "".join(f"  {t:8.2f} ({res[0] / t:4.2f}x)   " for t in res))
