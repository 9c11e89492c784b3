#!/usr/bin/env python3
"""The dense dL/dconf backward of the dual softmax (fm_dual_softmax_backward_dense: three tiled sweeps, no [N, L, S]
temporary) next to autograd through the reference's own expression (coarse_matching_new.py:64-68: einsum, two softmax,
product - three [N, L, S] temporaries and their gradients), at the bench's size.

    python tools/time_dsm_backward.py [N]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops, synth  # noqa: E402


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    l, c = 4800, 256
    f0n, f1n = synth.coarse_descriptors(77, 1, l, c, "borderline")
    f0 = torch.as_tensor(f0n, device=dev).repeat(n, 1, 1).contiguous()
    f1 = torch.as_tensor(f1n, device=dev).repeat(n, 1, 1).contiguous()
    g = torch.rand(n, l, l, device=dev)
    out = ops.coarse_match(f0, f1, (60, 80), (60, 80), 8.0, conf_matrix=True)
    buf = out['_coarse_buffers']
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ms_hip = timeit(lambda: ops._dsm_backward_dense(f0, f1, 0.1, buf, g))
    peak_hip = torch.cuda.max_memory_allocated() - base

    def ref():
        a0, a1 = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True)
        sim = torch.einsum("nlc,nsc->nls", a0, a1) / (c * 0.1)
        conf = torch.softmax(sim, 1) * torch.softmax(sim, 2)
        conf.backward(g)
        return a0.grad, a1.grad
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ms_ref = timeit(ref, 3)
    peak_ref = torch.cuda.max_memory_allocated() - base
    d0, d1 = ops._dsm_backward_dense(f0, f1, 0.1, buf, g)
    r0, r1 = ref()
    err = max((d0 - r0).abs().max().item() / r0.abs().max().item(), (d1 - r1).abs().max().item() / r1.abs().max().item())
    print(f"N={n} L=S={l} C={c}: HIP backward {ms_hip:.3f} ms (peak extra memory {peak_hip / 1e6:.0f} MB) | "
          f"torch forward+backward of the expression {ms_ref:.3f} ms (peak extra {peak_ref / 1e6:.0f} MB) | "
          f"relative difference of the gradients {err:.2e}")


if __name__ == "__main__":
    main()
