"""Every intermediate statistic of the coarse stage against numpy (tools/gpu_bringup.py as a test): the int8 planes
(|x - sigma q| <= sigma / 2 + clipped mass), row / column / unit maxima of the integer product (exact), stabilisers
(lower bounds of the true maxima), the partial sums, the candidate lists (superset of conf > thr) and the final ids -
at shapes that exercise the max pass's edge paths: a panel whose last wave holds only padding rows, a wave with one
real and one padding row block, fewer rows than a wave, C = 64 / 128 / 256, L != S, two samples."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hc0,wc0,hc1,wc1,c", [(8, 8, 8, 8, 64),        # L = S = 64: one wave's rows, no padding
                                               (5, 8, 6, 9, 64),        # L = 40 < 64, S = 54: padding inside a wave
                                               (12, 20, 12, 20, 128),   # L = 240: the panel's 4th wave is all padding
                                               (13, 17, 20, 30, 256),   # L = 221 (7 row blocks), S = 600, L != S
                                               (30, 40, 30, 40, 256)])  # L = 1200: 5 panels, the last one 176 rows
def test_every_coarse_statistic_against_numpy(hc0, wc0, hc1, wc1, c):
    from featurematching_amd import synth
    from tools import gpu_bringup
    f0, _ = synth.coarse_descriptors(7, 2, hc0 * wc0, c, "peaky")
    g0, g1 = synth.coarse_descriptors(7, 2, max(hc0 * wc0, hc1 * wc1), c, "peaky")
    f0, f1 = g0[:, :hc0 * wc0].copy(), g1[:, :hc1 * wc1].copy()
    assert gpu_bringup.run(f0, f1, (hc0, wc0), (hc1, wc1), label=f"{hc0}x{wc0} vs {hc1}x{wc1} C={c}")
