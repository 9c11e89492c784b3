#!/usr/bin/env python3
"""Same-process A/B of k_ctx_layer with 4 and with 8 waves per workgroup (a -DFM_TUNE_ENV build reads
FM_CTX_LAYER_WAVES at every call):  tools/build_variant.sh tune -DFM_TUNE_ENV && python tools/ab_ctx_waves.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib  # noqa: E402

_lib.load(os.path.join(ROOT, "build", "variants", "libfmatch_tune.so"))
from featurematching_amd import ops, synth  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    for n, l, layers in ((1, 4800, 8), (1, 4800, 2), (4, 4800, 8), (1, 1200, 8)):
        names = (['self', 'cross'] * layers)[:layers]
        wts = {k: torch.as_tensor(v) for k, v in synth.transformer_weights(77, 256, layers).items()}
        packed = ops.pack_coarse_transformer(wts, layers, dev)
        g = torch.Generator(device=dev).manual_seed(1)
        x0 = torch.randn(n, l, 256, device=dev, generator=g)
        x1 = torch.randn(n, l, 256, device=dev, generator=g)
        outs = {}
        for rep in range(2):
            for nw in ("4", "8"):
                os.environ["FM_CTX_LAYER_WAVES"] = nw
                us = timed(lambda: ops.coarse_transformer(x0, x1, packed, names))
                outs[nw] = ops.coarse_transformer(x0, x1, packed, names)
                print(f"N={n} L={l} layers={layers} rep {rep}: {nw} waves {us:8.1f} us", flush=True)
        d = max((outs["4"][0] - outs["8"][0]).abs().max().item(), (outs["4"][1] - outs["8"][1]).abs().max().item())
        print(f"   max |4 waves - 8 waves| = {d:.2e}", flush=True)


if __name__ == "__main__":
    main()
