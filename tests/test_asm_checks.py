"""Static checks of the generated gfx950 code (no GPU needed: hipcc cross-compiles here).

coarse_tf.hip issues its weight-fragment loads from inline asm with counted waits; hipcc believes the result of such
an asm is in its register at once, so any instruction it places on a ring register while the load is in flight would
read garbage - silently, and only when the load is late.  tools/check_inflight_regs.py walks the listing and proves
that no instruction touches the destination of an outstanding vector-memory load."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    src = os.path.join(ROOT, "featurematching_amd", "csrc", "coarse_tf.hip")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src), "-save-temps=obj", "-c", src, "-o", str(tmp_path / "coarse_tf.o")]
    subprocess.run(cmd, check=True, cwd=tmp_path, capture_output=True)
    listing = tmp_path / "coarse_tf-hip-amdgcn-amd-amdhsa-gfx950.s"
    assert listing.exists()
    for kernel in ("k_ctx_layer", "k_ctx_kvE"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py"), str(listing), kernel],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "instructions walked, 0 hazards" in r.stdout, r.stdout
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_the_checker_sees_a_register_touched_while_its_load_is_in_flight(tmp_path):
    """the checker itself: a copy of a loaded register before the wait is reported, the same copy after it is not"""
    listing = tmp_path / "toy.s"
    body = """_Z3toyv: ; @_Z3toyv
	global_load_dwordx4 v[4:7], v[0:1], off
	global_load_dwordx4 v[8:11], v[0:1], off offset:16
	%s
	s_waitcnt vmcnt(1)
	v_mov_b32_e32 v20, v4
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v21, v9
	s_endpgm
.Lfunc_end0:
"""
    tool = os.path.join(ROOT, "tools", "check_inflight_regs.py")
    listing.write_text(body % "v_mov_b32_e32 v12, v9")          # v9 is still in flight here
    r = subprocess.run([sys.executable, tool, str(listing), "toy"], capture_output=True, text=True)
    assert r.returncode == 1 and "touches in-flight registers [9]" in r.stdout, r.stdout
    listing.write_text(body % "v_mov_b32_e32 v12, v13")         # nothing in flight is touched
    r = subprocess.run([sys.executable, tool, str(listing), "toy"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 hazards" in r.stdout, r.stdout
