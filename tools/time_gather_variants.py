#!/usr/bin/env python3
"""Times the window-crop kernels of experimental builds (tools/build_variant.sh, e.g. the timing-only
-DFM_ABL_G_NOLOAD / -DFM_ABL_G_NOSTORE ablations) on a match list produced by the shipped library.

    python tools/time_gather_variants.py [--window 5] build/variants/libfmatch_X.so ...
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


def timed(fn, iters=30):
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    p = bench.Pair(bench.WORKLOADS["cfg2"], 1017, a.window, dev, "peaky")
    buf, _, _ = p.step()
    torch.cuda.synchronize()
    m = buf.read_count()
    c0, c1 = buf.cell_maps()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w, (hc, wc), (hf, wf) = a.window, p.hw_c, p.hw_f
    print(f"M={m} window={w}")
    for path in ["<shipped>"] + a.libs:
        v = _lib.load()
        if path != "<shipped>":
            v = C.CDLL(os.path.abspath(path))
            for name in ("fm_gather_windows", "fm_gather_windows_cells", "fm_fine_match"):
                res, args = _lib.ALL_SIGNATURES[name]
                getattr(v, name).restype, getattr(v, name).argtypes = res, args
        P = lambda t: C.c_void_p(t.data_ptr())
        out = []
        for ff, ids, cells, win in ((p.ff0, buf.i_ids, c0, p.win0), (p.ff1, buf.j_ids, c1, p.win1)):
            out.append(timed(lambda: v.fm_gather_windows(P(ff), 1, 64, hf, wf, 0, w, 4, 2, wc, P(buf.b_ids), P(ids),
                                                          P(buf.count), p.cap, P(win), st))[0])
            out.append(timed(lambda: v.fm_gather_windows_cells(P(ff), 1, 64, hf, wf, w, 4, 2, hc, wc, C.c_void_p(cells[0]),
                                                                cells[1], C.c_void_p(cells[2]), P(buf.b_ids), P(ids),
                                                                P(buf.count), p.cap, P(win), st))[0])
        k0 = torch.empty(p.cap, 3, device=dev); k1 = torch.empty(p.cap, 3, device=dev)
        tf = timed(lambda: v.fm_fine_match(P(p.win0), P(p.win1), p.cap, P(buf.count), w * w, 64, P(p.mix0), P(p.mix1),
                                           P(buf.mkpts0_c), P(buf.mkpts1_c), 2.0, P(k0), P(k1), st))[0]
        print(f"{os.path.basename(path):32s} fine {tf:6.1f} us")
        print(f"{os.path.basename(path):32s} img0 list {out[0]:6.1f}  cells {out[1]:6.1f}   img1 list {out[2]:6.1f}  cells {out[3]:6.1f} us")


if __name__ == "__main__":
    main()
