"""The C-ABI library: loads, exports every symbol include/fmatch.h declares, and validates
arguments on the host.  No compute call is made (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from featurematching_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(path=("include", "fmatch.h")):
    text = open(os.path.join(ROOT, *path)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    syms = declared_symbols()
    assert {"fm_coarse_match", "fm_gather_windows", "fm_fine_match", "fm_read_count",
            "fm_coarse_workspace_bytes", "fm_version", "fm_strerror"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in fmatch.h but not exported"
    assert set(_lib.SIGNATURES) == set(syms)
    # the public header is the boundary a maintainer of the reference reads: no diagnostics in it; those live in
    # csrc/fm_debug.h (bench.py / tools/ bracket single kernels with them) and are exported too
    assert not [s for s in syms if s.startswith("fm_debug_")]
    dbg = declared_symbols(("featurematching_amd", "csrc", "fm_debug.h"))
    assert dbg and all(s.startswith("fm_debug_") for s in dbg) and set(_lib.DEBUG_SIGNATURES) == set(dbg)
    for s in dbg:
        assert hasattr(lib, s), f"{s} declared in fm_debug.h but not exported"


def test_version_and_messages():
    lib = _lib.load()
    assert lib.fm_version() == 100
    assert lib.fm_strerror(0) == b"ok"
    for code in range(-10, 0):
        assert lib.fm_strerror(code) not in (b"", b"unknown fmatch status")
    assert lib.fm_default_cand_slots(0.2) == 8
    assert lib.fm_default_cand_slots(0.05) == 32
    assert lib.fm_default_cand_slots(0.01) == 64


def test_workspace_query_and_argument_checks():
    lib = _lib.load()
    n = C.c_size_t(0)
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 8, C.byref(n)) == 0
    per_pair = n.value
    assert 10e6 < per_pair < 60e6          # any mode: 4 float16 planes (9.96 MB) + both candidate sets + statistics
    # the common path (mode 0: prep, max pass, sparse sum kernel, assignment) needs neither the float16 planes nor the
    # dense kernel's partials and candidate set
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 0, 0, C.byref(n)) == 0
    assert 2.4e6 < n.value <= 8e6          # 2 int8 planes (2.47 MB) + statistics, partial sums, candidate lists
    common = n.value
    for mode, conf in ((1, 0), (2, 0), (3, 0), (0, 1)):
        assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, mode, conf, C.byref(n)) == 0
        assert n.value == per_pair > common
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 128, 0, C.byref(n)) == -3    # unknown mode bit
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 64, 0, C.byref(n)) == 0 and n.value == common   # FM_MODE_ALONE: geometry only
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 8, 0, C.byref(n)) == 0 and n.value == common   # FM_MODE_EXACT_STEP
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 4, 0, C.byref(n)) == 0 and n.value == common   # FM_MODE_NO_CELL_MAPS
    assert lib.fm_coarse_workspace_bytes_mode(1, 4800, 4800, 256, 8, 0, 0, None) == -1
    assert lib.fm_coarse_workspace_bytes(64, 4800, 4800, 256, 8, C.byref(n)) == 0
    assert n.value < 64 * per_pair * 1.2
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 102, 8, C.byref(n)) == -3     # C % 4 != 0
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 512, 8, C.byref(n)) == -3     # C > 256
    assert lib.fm_coarse_workspace_bytes(1, 64, 64, 32, 8, C.byref(n)) == 0           # zero-padded to 64
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 7, C.byref(n)) == -3     # slots not a power of 2
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 2, C.byref(n)) == -3     # fewer than 4 slots
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 128, C.byref(n)) == -3   # more than 64
    assert lib.fm_coarse_workspace_bytes(0, 4800, 4800, 256, 8, C.byref(n)) == -2
    assert lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 8, None) == -1
    # NULL / shape checks return before any device work
    null = None
    args = [null, null, 1, 64, 64, 64, 8, 8, 8, 8, 0.1, 0.2, 2, 8.0, null, null, null, 0, 8, 0,
            null, null, null, null, null, null, 0, null, null, null]
    assert lib.fm_coarse_match(*args) == -1
    one_ = C.c_void_p(256)
    dargs = [one_, one_, 3, 1, 64, 64, 64, 8, 8, 8, 8, 0.1, 0.2, 2, 8.0, null, null, one_, 1 << 30, 8, 0,
             one_, one_, one_, one_, one_, one_, 64, one_, null, null]
    assert lib.fm_coarse_match_dtype(*dargs) == -3                 # unknown element type
    assert lib.fm_gather_windows(null, 1, 64, 8, 8, 0, 7, 4, 2, 2, null, null, null, 4, null, null) == -1
    assert lib.fm_gather_windows(null, 1, 64, 8, 8, 0, 7, 4, 2, 2, null, null, null, 0, null, null) == 0   # M == 0
    assert lib.fm_fine_match(null, null, 0, null, 49, 64, null, null, null, null, 2.0, null, null, null) == 0
    assert lib.fm_fine_match(null, null, 3, null, 49, 64, null, null, null, null, 2.0, null, null, null) == -1
    # cell-ordered and merging crops: M == 0 is a no-op, NULL pointers are refused, the shape limits hold
    one = C.c_void_p(256)        # a non-NULL address that is never dereferenced on these paths
    assert lib.fm_gather_windows_cells(null, 1, 64, 8, 8, 7, 4, 2, 2, 2, null, 4, null, null, null, null, 0, null, null) == 0
    assert lib.fm_gather_windows_cells(null, 1, 64, 8, 8, 7, 4, 2, 2, 2, null, 4, null, null, null, null, 3, null, null) == -1
    assert lib.fm_gather_windows_cells(one, 1, 64, 8, 8, 7, 4, 2, 2, 2, one, 3, one, one, one, null, 3, one, null) == -2  # pitch < cells
    assert lib.fm_gather_windows_cells(one, 1, 32, 8, 8, 7, 4, 2, 2, 2, one, 4, one, one, one, null, 3, one, null) == -3  # Cf != 64
    assert lib.fm_gather_merge_windows(null, 1, 64, 8, 8, 7, 4, 2, 2, 2, null, 0, null, null, null, null, null, null, 0,
                                       null, null) == 0
    assert lib.fm_gather_merge_windows(one, 1, 64, 8, 8, 7, 4, 2, 2, 2, null, 0, null, null, one, one, one, null, 3,
                                       one, null) == -1          # packed weights missing
    assert lib.fm_gather_merge_windows(one, 1, 64, 8, 8, 7, 4, 2, 2, 2, one, 4, null, one, one, one, one, null, 3,
                                       one, null) == -1          # cell map without its tie list
    assert lib.fm_gather_merge_windows(one, 1, 64, 8, 8, 9, 4, 2, 2, 2, null, 0, null, one, one, one, one, null, 3,
                                       one, null) == -3          # W not in {5,7}
    assert lib.fm_merge_pack_weights(null, 64, null, null) == -1
    assert lib.fm_merge_pack_weights(one, 32, one, null) == -3


def test_debug_entry_points_validate_their_shapes():
    """The diagnostic entry points make the same N/L/S/C/slots checks as the product ones (a zero N used to reach
    an integer division on the host)."""
    lib = _lib.load()
    one = C.c_void_p(256)
    arr = (C.c_int64 * 40)()
    assert lib.fm_debug_coarse_layout(0, 64, 64, 64, 8, arr, 40) == -2
    assert lib.fm_debug_coarse_layout(1, 64, 64, 66, 8, arr, 40) == -3
    assert lib.fm_debug_coarse_layout(1, 64, 64, 64, 3, arr, 40) == -3
    assert lib.fm_debug_coarse_layout(1, 64, 64, 64, 8, arr, 39) == -2
    assert lib.fm_debug_reset_counters(one, 0, 64, 64, 64, 8, None) == -2
    assert lib.fm_debug_reset_counters(one, 1, 64, 64, 64, 5, None) == -3
    assert lib.fm_debug_reset_counters(None, 1, 64, 64, 64, 8, None) == -1
    assert lib.fm_debug_launch_corr(one, 1, 0, 64, 64, 8, 0.1, 0.2, 1, None) == -2
    assert lib.fm_debug_launch_corr(one, 1, 64, 64, 64, 8, 0.1, 0.2, 7, None) == -3
    assert lib.fm_debug_launch_screen(one, one, one, 1, 64, -1, 64, 8, 0.1, 0.2, None) == -2
    assert lib.fm_debug_launch_screen(one, None, one, 1, 64, 64, 64, 8, 0.1, 0.2, None) == -1


def test_layout_query_is_consistent():
    lib = _lib.load()
    arr = (C.c_int64 * 40)()
    assert lib.fm_debug_coarse_layout(1, 4800, 4800, 256, 8, arr, 40) == 0
    v = list(arr)
    assert v[:4] == [1, 4800, 4800, 256]
    assert v[4] == 4864 and v[5] == 4800 and v[6] == 19 and v[7] == 75      # Lp, Sp, panels, tiles
    assert 1 <= v[8] <= 32 and v[6] * v[8] <= 256                            # one round of the 256 CUs
    offs = v[10:37] + v[39:]
    assert all(o % 256 == 0 for o in offs)
    # screening kernel: chunks x units per chunk cover Sp/32, at most 64 units per chunk (one ballot per chunk)
    assert v[37] * v[38] >= 150 and v[38] in (32, 64) and (v[37] - 1) * v[38] < 150
    n = C.c_size_t(0)
    lib.fm_coarse_workspace_bytes(1, 4800, 4800, 256, 8, C.byref(n))
    assert offs[-1] == n.value


def test_ops_refuse_cpu_tensors():
    import torch
    from featurematching_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.coarse_match(torch.zeros(1, 64, 64), torch.zeros(1, 64, 64), (8, 8), (8, 8), 8.0)
    # the context-layer kernels likewise: the module keeps its torch ops for CPU tensors, the ops themselves refuse
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.coarse_transformer(torch.zeros(1, 40, 256), torch.zeros(1, 40, 256), torch.zeros(16, dtype=torch.uint8), ['self'])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fine_transformer(torch.zeros(3, 49, 64), torch.zeros(3, 49, 64), torch.zeros(16, dtype=torch.uint8))
    from featurematching_amd.transformer import LocalFeatureTransformer
    tf = LocalFeatureTransformer(dict(d_model=256, nhead=8, layer_names=['self', 'cross'], attention='linear')).eval()
    with torch.no_grad():
        a0, a1 = tf(torch.randn(1, 40, 256), torch.randn(1, 33, 256))       # CPU tensors: the torch layers
    assert a0.shape == (1, 40, 256) and a1.shape == (1, 33, 256) and tf._hip_kind(a0, a1) is None


def test_c_driver_under_address_sanitizer():
    """tests/c/abi_driver.c - a plain C caller of the ABI - against a build of the library whose HOST code is
    instrumented with AddressSanitizer (make asan; the device code is untouched, GPU ASan is not available on the
    pool): every host-decided status < 0 of fmatch.h and the workspace-layout arithmetic over ragged shapes."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "featurematching_amd", "csrc")
    if not (shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc toolchain on this machine")
    subprocess.run(["make", "-C", csrc, "asan", "-j6"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([os.path.join(ROOT, "build", "asan", "abi_driver"),
                        os.path.join(ROOT, "build", "asan", "libfmatch_hip_asan.so")],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr
