#!/usr/bin/env python3
"""In-kernel shader clock of the dense conf_matrix sweep at the batch of 64 pairs (diagnostic build:
tools/build_variant.sh diagclock -DFM_DIAG_CLOCK): per wave of split 0 of the first 8 panels of sample 0 the cycles in
the kernel, in the MFMA chains (+ epilogue slices), at the per-unit barrier, in the prologue.

    python tools/diag_conf_clock.py build/variants/libfmatch_diagclock.so [N]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib  # noqa: E402


def main():
    _lib.load(os.path.abspath(sys.argv[1]))
    import bench
    from tools.gpu_bringup import layout
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    p = bench.Pair(dict(bench.WORKLOADS["cfg3"], n=n), 4242, 5, dev, "peaky", device_data=True)
    p.conf_matrix, p.dense, p.stages, p.fuse_maps = True, True, "coarse", False
    for _ in range(4):
        p.step()
    torch.cuda.synchronize()
    buf = p.last[0]
    ws = buf.workspace
    off = (-ws.data_ptr()) % 256
    slots = p.slots or _lib.load().fm_default_cand_slots(0.2)
    lay = layout(p.n, p.l, p.l, p.c, slots)
    o = off + lay["cand_x"] + p.l * slots * 4
    d = ws[o: o + 512 * 4].view(torch.float32).cpu().numpy().reshape(64, 8)
    names = ["cycles", "ticks100MHz", "units", "chain_cyc", "-", "barrier_cyc", "prologue_cyc", "tail_cyc"]
    print(f"N={n}: {lay['splits']} splits; clock {np.median(d[:, 0]) / np.median(d[:, 1]) * 0.1:.2f} GHz")
    for k, nme in enumerate(names):
        col = d[:, k]
        print(f"   {nme:14s} median {np.median(col):9.0f}  mean {col.mean():9.0f}  min {col.min():9.0f}  max {col.max():9.0f}")
    u = d[:, 2].sum()
    print(f"   per unit: chain+epilogue {d[:, 3].sum() / u:.0f} cyc, barrier {d[:, 5].sum() / u:.0f} cyc, all {d[:, 0].sum() / u:.0f} cyc")


if __name__ == "__main__":
    main()
