#!/usr/bin/env python3
"""Time fm_coarse_transformer against the PyTorch-ROCm module on [N, L, 256] token sets (default: one 640x480 pair,
the reference's 8 layers)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops, synth  # noqa: E402
from featurematching_amd.transformer import LocalFeatureTransformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--l", type=int, default=4800)
    ap.add_argument("--layers", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = (['self', 'cross'] * a.layers)[:a.layers]
    wts = {k: torch.as_tensor(v) for k, v in synth.transformer_weights(77, 256, a.layers).items()}
    tf = LocalFeatureTransformer(dict(d_model=256, nhead=8, layer_names=names, attention='linear')).to(dev).eval()
    tf.load_state_dict(wts)
    packed = ops.pack_coarse_transformer(wts, a.layers, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(a.n, a.l, 256, device=dev, generator=g)
    x1 = torch.randn(a.n, a.l, 256, device=dev, generator=g)
    tf.use_hip = False                          # the module itself takes the HIP kernels in eval mode: keep its torch ops
    with torch.no_grad():
        r0, r1 = tf(x0, x1)
    h0, h1 = ops.coarse_transformer(x0, x1, packed, names)
    print(f"N={a.n} L={a.l} layers={a.layers}: max |HIP - torch| = {(h0 - r0).abs().max().item():.2e} / "
          f"{(h1 - r1).abs().max().item():.2e}  (|out| max {r0.abs().max().item():.1f})")
    flop = 2.0 * a.n * 2 * a.l * 655360 * a.layers
    for name, fn in (("HIP fm_coarse_transformer", lambda: ops.coarse_transformer(x0, x1, packed, names)),
                     ("torch module", lambda: tf(x0, x1))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"   {name:28s} {ms * 1e3:9.1f} us   ({flop / ms / 1e9:.1f} TFLOP/s of float32 work)")


if __name__ == "__main__":
    main()
