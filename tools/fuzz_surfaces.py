#!/usr/bin/env python3
"""Randomised shapes through the parametrised GPU parity tests of the surfaces either side of the matching path (not part
of the suite): the coarse context layers (test_coarse_transformer_vs_oracle) and the dual-softmax backward for a dense
dL/dconf (test_dense_conf_matrix_gradient_goes_through_the_hip_backward).

    python tools/fuzz_surfaces.py [--cases 40] [--seed 0]
"""
import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        n, l, s = int(rng.integers(1, 4)), int(rng.integers(33, 700)), int(rng.integers(33, 700))
        layers = [str(rng.choice(['self', 'cross'])) for _ in range(int(rng.integers(1, 5)))]
        hw0 = (int(rng.integers(5, 21)), int(rng.integers(5, 21)))
        hw1 = (int(rng.integers(5, 21)), int(rng.integers(5, 21)))
        c = int(rng.choice([32, 64, 128, 256]))
        loss = str(rng.choice(["focal", "cross_entropy", "weighted_sum"]))
        for name, fn, args in (("coarse_transformer", T.test_coarse_transformer_vs_oracle, (n, l, s, layers)),
                               ("dense_backward", T.test_dense_conf_matrix_gradient_goes_through_the_hip_backward, (hw0, hw1, c, loss))):
            try:
                fn(*args)
                print(f"ok   case {case:3d} {name} {args}", flush=True)
            except Exception:
                bad += 1
                print(f"FAIL case {case:3d} {name} {args}\n" + traceback.format_exc(limit=3), flush=True)
    print(f"{2 * a.cases - bad} / {2 * a.cases} runs pass")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
