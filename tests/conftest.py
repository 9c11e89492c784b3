import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Guard-band flips of every match-set comparison against a fixture / the oracle (the parity tests tolerate a
    reference / HIP disagreement only for entries whose conf lies within the guard band around thr): how many actually
    occurred."""
    try:
        from helpers import FLIPS
    except Exception:
        return
    if not FLIPS:
        return
    tot = sum(f for _, f, _, _ in FLIPS)
    terminalreporter.write_line(f"guard-band flips: {tot} in {len(FLIPS)} match-set comparisons "
                                f"({sum(n for _, _, n, _ in FLIPS)} reference matches); comparisons with flips:")
    for name, f, n, err in FLIPS:
        if f:
            terminalreporter.write_line(f"  {name}: {f} of {n} (max |conf - ref| {err:.2e})")
