#!/usr/bin/env python3
"""Whose float32 noise is it?  mconf of the HIP path and of the float32 oracle (torch-CPU, the reference's ops) against
a float64 evaluation of coarse_matching_new.py:64-68 on the same inputs.

    python tools/diag_conf_f64.py [--workload cfg5] [--dist borderline]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import ops, synth  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (checker only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg5")
    ap.add_argument("--dist", default="borderline")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.workload]
    sh = synth.config_shapes(cfg)
    f0, f1 = synth.coarse_descriptors(cfg["seed"], 1, sh["l"], cfg["c"], a.dist)
    dev = torch.device("cuda:0")
    t0, t1 = torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev)
    hw_c = (sh["hc"], sh["wc"])
    out = ops.coarse_match(t0, t1, hw_c, hw_c, 8.0)
    sim = (t0[0].double() @ t1[0].double().T) / (cfg["c"] * 0.1)
    conf64 = torch.softmax(sim, 0) * torch.softmax(sim, 1)
    i, j = out["i_ids"], out["j_ids"]
    truth = conf64[i, j]
    e_hip = (out["mconf"].double() - truth).abs()
    torch.set_num_threads(16)
    ref = orc.coarse_match(f0, f1, (cfg["h"], cfg["w"]), hw_c, hw_c, 0.2, 2, 0.1)
    ri, rj = ref["i_ids"].to(dev), ref["j_ids"].to(dev)
    e_ref = (ref["mconf"].to(dev).double() - conf64[ri, rj]).abs()
    print(f"{a.workload} {a.dist}: M = {i.numel()} (oracle {ri.numel()})")
    print(f"  HIP    vs float64: max {e_hip.max().item():.3e}  mean {e_hip.mean().item():.3e}")
    print(f"  oracle vs float64: max {e_ref.max().item():.3e}  mean {e_ref.mean().item():.3e}")
    if i.numel() == ri.numel() and bool((i == ri).all()) and bool((j == rj).all()):
        d = (out["mconf"].double() - ref["mconf"].to(dev).double()).abs()
        k = int(d.argmax())
        print(f"  HIP vs oracle: max {d.max().item():.3e} at match {k}: conf {truth[k].item():.6f}, HIP err "
              f"{(out['mconf'][k].double() - truth[k]).item():+.3e}, oracle err {(ref['mconf'][k].to(dev).double() - truth[k]).item():+.3e}, "
              f"sim {sim[i[k], j[k]].item():.3f}")


if __name__ == "__main__":
    main()
