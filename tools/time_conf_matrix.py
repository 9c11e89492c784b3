#!/usr/bin/env python3
"""The coarse stage with data['conf_matrix'] requested at the batch of 64 pairs (BASELINE config 3's HBM-bound mode: the
5.9 GB float32 write), a few steps - run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    if '--lib' in sys.argv:            # an experimental build (tools/build_variant.sh), loaded explicitly
        at = sys.argv.index('--lib')
        from featurematching_amd import _lib
        _lib.load(os.path.abspath(sys.argv[at + 1]))
        del sys.argv[at:at + 2]
    dist = 'peaky'
    if '--dist' in sys.argv:           # 'borderline': every sample goes through the dense kernel and the hi/lo-split sweep
        at = sys.argv.index('--dist')
        dist = sys.argv[at + 1]
        del sys.argv[at:at + 2]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    wl = dict(bench.WORKLOADS["cfg3"], n=n)
    p = bench.Pair(wl, 4242, 5, dev, dist, device_data=True)
    if dist != "peaky":
        p.slots = 16
    p.conf_matrix, p.dense, p.stages, p.fuse_maps = True, True, "coarse", False
    p.step()
    p.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nbytes = 4.0 * n * p.l * p.l
    print(f"N={n}: coarse stage with conf_matrix {ms:.3f} ms per step = {nbytes / ms / 1e6:.1f} GB/s of conf_matrix bytes over the whole stage")


if __name__ == "__main__":
    main()
