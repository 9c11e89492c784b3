#!/usr/bin/env python3
"""Record the match lists the HIP path produces for a small batch (4 pairs of BASELINE config #1: 128x128, C=64)
as tests/golden/hip_matches_cfg1x4.npz.  The world-size-2 gloo test (tests/test_dist.py) feeds these RECORDED HIP
outputs through pack -> gather -> unpack and compares with the single-process concatenation; the file also holds
the oracle's lists for the same inputs, so the recording itself is checked when it is made.

    python tools/record_hip_matches.py        (on the GPU box; writes the .npz next to the other fixtures)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops, synth  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (checker only)


def main():
    cfg = dict(synth.CONFIGS["cfg1"], n=4)
    sh = synth.config_shapes(cfg)
    seed = 41
    f0, f1 = synth.coarse_descriptors(seed, 4, sh["l"], cfg["c"], "peaky")
    ff0, ff1 = synth.fine_maps(seed, 4, cfg["cf"], sh["hf"], sh["wf"])
    mix = synth.mix_weights(seed, 49)
    hw_i, hw_c = (cfg["h"], cfg["w"]), (sh["hc"], sh["wc"])
    dev = torch.device("cuda:0")
    out = ops.coarse_match(torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev), hw_c, hw_c, hw_i[0] / hw_c[0])
    win0 = ops.gather_windows(torch.as_tensor(ff0, device=dev), out["b_ids"], out["i_ids"], 7, 4, hw_c[1])
    win1 = ops.gather_windows(torch.as_tensor(ff1, device=dev), out["b_ids"], out["j_ids"], 7, 4, hw_c[1])
    mix0 = torch.as_tensor(np.concatenate([mix[0], [mix[1]]]).astype(np.float32), device=dev)
    mix1 = torch.as_tensor(np.concatenate([mix[2], [mix[3]]]).astype(np.float32), device=dev)
    k0, k1 = ops.fine_match(win0, win1, mix0, mix1, out["mkpts0_c"], out["mkpts1_c"], hw_i[0] / sh["hf"])
    ref = orc.match_features(f0, f1, ff0, ff1, hw_i, mix, w=7)
    b = out["b_ids"].cpu().numpy()
    assert np.array_equal(b, ref["b_ids"].numpy()) and np.array_equal(out["i_ids"].cpu().numpy(), ref["i_ids"].numpy())
    assert np.abs(k0.cpu().numpy() - ref["mkpts0_f"].numpy()).max() < 1e-3
    path = os.path.join(ROOT, "tests", "golden", "hip_matches_cfg1x4.npz")
    np.savez_compressed(path, b_ids=b.astype(np.int32), kpts0=k0.cpu().numpy()[:, :2], kpts1=k1.cpu().numpy()[:, :2],
                        mconf=out["mconf"].cpu().numpy(), seed=np.int32(seed))
    print("wrote", path, "M =", b.shape[0], "per pair", np.bincount(b, minlength=4).tolist())


if __name__ == "__main__":
    main()
