#!/usr/bin/env python3
"""Run bench.py against an experimental build of the library (tools/build_variant.sh NAME "-D..."):

    python tools/bench_variant.py build/variants/libfmatch_NAME.so [bench.py arguments]

The product's loader takes no path from the environment; this tool loads the variant first, explicitly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402,F401
from featurematching_amd import _lib  # noqa: E402

if __name__ == "__main__":
    _lib.load(os.path.abspath(sys.argv[1]))
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    import bench  # noqa: E402
    bench.main()
