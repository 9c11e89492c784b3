#!/usr/bin/env python3
"""How many torch streams really run side by side: S streams each replay a graph of 6 spin kernels (torch.cuda._sleep);
with full concurrency the time per kernel per stream stays at the 1-stream value.  HIP maps streams onto a few hardware
queues; two streams on one queue serialise."""
import sys
import time

import torch

dev = torch.device("cuda:0")
torch.cuda._sleep(1000)
torch.cuda.synchronize()
K, R, CYC = 6, 150, 40000


def trial(streams, label):
    graphs = []
    for st in streams:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(K):
                torch.cuda._sleep(CYC)
        graphs.append(g)
    for _ in range(10):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6
    print(f"{label:34s} {len(streams)} streams: {us / (R * K * len(streams)):6.2f} us per kernel (job)  "
          f"{us / (R * K):6.2f} per kernel per stream")


pool = [torch.cuda.Stream(dev) for _ in range(12)]
for s in (1, 2, 3, 4, 5, 6, 8):
    trial(pool[:s], "pool[0:S]")
trial(pool[1:5], "pool[1:5]")
trial(pool[2:6], "pool[2:6]")
trial([pool[0], pool[1], pool[2], pool[4]], "pool[0,1,2,4]")
trial([pool[0], pool[1], pool[2], pool[5]], "pool[0,1,2,5]")
hi = [torch.cuda.Stream(dev, priority=-1) for _ in range(4)]
trial(hi[:3], "high priority x3")
trial(hi[:4], "high priority x4")
trial(pool[:3] + hi[:1], "3 normal + 1 high")
trial(pool[:3] + hi[:3], "3 normal + 3 high")
trial(pool[:2] + hi[:2], "2 normal + 2 high")
