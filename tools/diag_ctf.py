#!/usr/bin/env python3
"""In-kernel shader-clock stamps of k_ctx_layer (diagnostic build: tools/build_variant.sh diagctf -DFM_DIAG_CTF):
per wave, cycles from kernel entry to the end of each stage of the LAST layer launch.

    python tools/diag_ctf.py build/variants/libfmatch_diagctf.so [--layers 1]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops, synth  # noqa: E402

STAGES = ["x tile", "q gemm", "attention", "barrier", "merge gemm", "LN1+store+bar", "mlp1 gemm", "store+bar",
          "mlp2 gemm", "LN2", "store out"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--l", type=int, default=4800)
    ap.add_argument("--layers", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = (['self', 'cross'] * a.layers)[:a.layers]
    wts = {k: torch.as_tensor(v) for k, v in synth.transformer_weights(77, 256, a.layers).items()}
    packed = ops.pack_coarse_transformer(wts, a.layers, dev)
    v = C.CDLL(os.path.abspath(a.lib))
    for name in ("fm_coarse_tf_workspace_bytes", "fm_coarse_transformer"):
        getattr(v, name).restype, getattr(v, name).argtypes = _lib.ALL_SIGNATURES[name]
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(1, a.l, 256, device=dev, generator=g)
    x1 = torch.randn(1, a.l, 256, device=dev, generator=g)
    nb = C.c_size_t()
    v.fm_coarse_tf_workspace_bytes(1, a.l, a.l, C.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    o0, o1 = torch.empty_like(x0), torch.empty_like(x1)
    kinds = (C.c_int * a.layers)(*[{'self': 0, 'cross': 1}[k] for k in names])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = v.fm_coarse_transformer(x0.data_ptr(), x1.data_ptr(), 1, a.l, a.l, 256, 8, kinds, a.layers, packed.data_ptr(),
                                     ws.data_ptr(), nb.value, o0.data_ptr(), o1.data_ptr(), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    tiles = (a.l + 31) // 32
    nwg = 2 * tiles if names[-1] == 'self' else tiles
    d = ws.view(torch.float32)[-2 * tiles * 64:][: nwg * 64].cpu().numpy().reshape(nwg * 4, 16)
    print(f"{nwg} workgroups x 4 waves, last layer '{names[-1]}'; cumulative cycles (median / min / max over waves), step")
    prev = 0.0
    for k, nme in enumerate(STAGES):
        col = d[:, k]
        print(f"   {nme:14s} {np.median(col):9.0f} {col.min():9.0f} {col.max():9.0f}   +{np.median(col) - prev:8.0f}")
        prev = np.median(col)


if __name__ == "__main__":
    main()
