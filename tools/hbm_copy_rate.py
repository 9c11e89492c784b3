#!/usr/bin/env python3
"""Practical HBM ceiling for a read+write stream on this GPU: torch's device-to-device copy at several sizes (the crop
and fine kernels are priced against the 8 TB/s datasheet peak; this is what a plain copy reaches)."""
import torch

dev = torch.device("cuda:0")
for mb in (48, 96, 512, 2048):
    n = mb * 1024 * 1024 // 4
    srcs = [torch.randn(n, device=dev) for _ in range(max(2, 1024 // mb))]      # cycle through > Infinity Cache
    dst = torch.empty(n, device=dev)
    for s in srcs[:2]:
        dst.copy_(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(8, 4096 // mb)
    e0.record()
    for i in range(iters):
        dst.copy_(srcs[i % len(srcs)])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"copy {mb:5d} MB (read) + {mb} MB (write): {ms * 1e3:8.1f} us  -> {2 * mb * 1.048576 / ms:7.1f} GB/s read+write")
