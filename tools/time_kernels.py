#!/usr/bin/env python3
"""Per-stage timing of the hot path on one GPU (events on the launch stream, median of --iters).

    python tools/time_kernels.py [--workload cfg2] [--window 5] [--iters 30] [--layout nchw|nhwc]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops, synth  # noqa: E402
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
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dist", default="peaky")
    ap.add_argument("--layout", default="nchw")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[a.workload]
    p = bench.Pair(wl, 1017, a.window, dev, a.dist, fine_path="windows")   # this tool times the window-tensor kernels
    if a.layout == "nhwc":
        p.ff0 = p.ff0.contiguous(memory_format=torch.channels_last)
        p.ff1 = p.ff1.contiguous(memory_format=torch.channels_last)
    lib = _lib.load()
    buf, k0, k1 = p.step()
    torch.cuda.synchronize()
    m = buf.read_count()
    print(f"workload {a.workload}: N={p.n} L={p.l} C={p.c} window={a.window} M={m}")
    ws = buf.workspace
    ptr = C.c_void_p(ws.data_ptr() + ((-ws.data_ptr()) % 256))
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = a.window
    rows = []

    def reset():
        lib.fm_debug_reset_counters(ptr, p.n, p.l, p.l, p.c, slots, st)

    rows.append(("coarse (whole fm_coarse_match)",
                 timed(lambda: ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap), a.iters)))
    if p.n * p.l * p.l * 4 <= 8e9:      # dense conf_matrix (training surface): one more sweep + N*L*S*4 bytes written
        rows.append(("coarse + dense conf_matrix (k_dense<., CONF>)",
                     timed(lambda: ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, conf_matrix=True),
                           max(3, a.iters // 4))))
    rows.append(("  corr max pass (k_max_i8)",
                 timed(lambda: lib.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 0, st), a.iters)))
    rows.append(("  corr sum pass (k_dense)",
                 timed(lambda: lib.fm_debug_launch_corr(ptr, p.n, p.l, p.l, p.c, slots, 0.1, 0.2, 1, st), a.iters, reset)))
    p.step()                        # the sweep timings above reset this workspace
    buf = ops.coarse_match_async(p.f0, p.f1, p.hw_c, p.hw_c, 8.0, cap=p.cap, cell_maps=True)   # windows path: with cell maps
    torch.cuda.synchronize()
    rows.append(("gather windows (both images, list order)", timed(lambda: (
        ops.gather_windows(p.ff0, buf.b_ids, buf.i_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win0),
        ops.gather_windows(p.ff1, buf.b_ids, buf.j_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win1)), a.iters)))
    rows.append(("  gather image 0 (sorted cells)", timed(lambda: ops.gather_windows(
        p.ff0, buf.b_ids, buf.i_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win0), a.iters)))
    rows.append(("  gather image 1 (permuted cells)", timed(lambda: ops.gather_windows(
        p.ff1, buf.b_ids, buf.j_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win1), a.iters)))
    c0, c1 = buf.cell_maps()
    rows.append(("  gather image 0, cell order", timed(lambda: ops.gather_windows(
        p.ff0, buf.b_ids, buf.i_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win0, cells=c0, h_c=p.hw_c[0]), a.iters)))
    rows.append(("  gather image 1, cell order", timed(lambda: ops.gather_windows(
        p.ff1, buf.b_ids, buf.j_ids, w, 4, p.hw_c[1], count=buf.count, out=p.win1, cells=c1, h_c=p.hw_c[0]), a.iters)))
    mw = torch.as_tensor(synth.merge_weights(7, p.c, 64)[2], device=dev)
    packed = ops.pack_merge_weights(mw)
    ctx = torch.randn(p.n, p.l, 64, device=dev)
    rows.append(("  crop + context merge, image 0", timed(lambda: ops.gather_merge_windows(
        p.ff0, packed, ctx, buf.b_ids, buf.i_ids, w, 4, p.hw_c[0], p.hw_c[1], count=buf.count, out=p.win0, cells=c0), a.iters)))
    rows.append(("  crop + context merge, image 1", timed(lambda: ops.gather_merge_windows(
        p.ff1, packed, ctx, buf.b_ids, buf.j_ids, w, 4, p.hw_c[0], p.hw_c[1], count=buf.count, out=p.win1, cells=c1), a.iters)))
    rows.append(("fine match", timed(lambda: ops.fine_match(p.win0, p.win1, p.mix0, p.mix1, buf.mkpts0_c, buf.mkpts1_c,
                                                              2.0, count=buf.count), a.iters)))
    rows.append(("whole step (eager)", timed(lambda: p.step(), a.iters)))
    flops = 2.0 * p.n * p.l * p.l * p.c
    for name, (med, mn) in rows:
        extra = ""
        if "corr" in name:
            extra = f"   {flops / (med * 1e-6) / 1e12:8.1f} TFLOP/s algorithmic"
        print(f"{name:36s} median {med:9.1f} us   min {mn:9.1f} us{extra}")
    t = dict((n, v[0]) for n, v in rows)
    if "coarse + dense conf_matrix (k_dense<., CONF>)" in t:
        dt = t["coarse + dense conf_matrix (k_dense<., CONF>)"] - t["coarse (whole fm_coarse_match)"]
        gb = p.n * p.l * p.l * 4 / 1e9
        print(f"dense conf_matrix: {gb:.2f} GB written in {dt:.0f} us more than the fused path -> {gb / (dt * 1e-6) / 1e3:.2f} TB/s")
    wbytes = 2 * m * w * w * 64 * 4
    print(f"window bytes (both images): {wbytes / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
