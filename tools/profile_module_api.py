#!/usr/bin/env python3
"""Host-side profile (cProfile) of the drop-in module path: CoarseMatching -> window crop -> FineMatching, one 640x480
pair per call."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
print("pairs/s:", bench.module_api_rate(wl, 5, dev, iters=100))
pr = cProfile.Profile()
pr.enable()
bench.module_api_rate(wl, 5, dev, iters=200)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
