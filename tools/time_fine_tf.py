#!/usr/bin/env python3
"""Time fm_fine_transformer against the PyTorch-ROCm module on M windows (default: the cfg#2 match count)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops, synth  # noqa: E402
from featurematching_amd.transformer import LocalFeatureTransformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=3800)
    ap.add_argument("--w", type=int, default=7)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ww = a.w * a.w
    wts = {k: torch.as_tensor(v) for k, v in synth.transformer_weights(77, 64, 2).items()}
    tf = LocalFeatureTransformer(dict(d_model=64, nhead=8, layer_names=['self', 'cross'], attention='linear')).to(dev).eval()
    tf.load_state_dict(wts)
    packed = ops.pack_fine_transformer(wts, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(a.m, ww, 64, device=dev, generator=g)
    x1 = torch.randn(a.m, ww, 64, device=dev, generator=g)
    tf.use_hip = False                          # the module itself takes the HIP kernel in eval mode: keep its torch ops
    with torch.no_grad():
        r0, r1 = tf(x0, x1)
    h0, h1 = ops.fine_transformer(x0, x1, packed)
    print(f"M={a.m} W={a.w}: max |HIP - torch| = {(h0 - r0).abs().max().item():.2e} / {(h1 - r1).abs().max().item():.2e}")
    for name, fn in (("HIP fm_fine_transformer", lambda: ops.fine_transformer(x0, x1, packed)),
                     ("torch module", lambda: tf(x0, x1))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        flop = a.m * 4 * 2.0 * ww * (4 * 64 * 64 + 128 * 128 + 128 * 64 + 2 * 64 * 8)
        print(f"   {name:26s} {ms * 1e3:9.1f} us   ({flop / ms / 1e9:.1f} TFLOP/s of float32 work)")


if __name__ == "__main__":
    main()
