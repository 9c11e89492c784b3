"""The match-list exchange (the reference's utils/comm.py:113-176 scheme) on RCCL: a world of ONE rank on
the one GPU of the test box - `init_process_group("nccl", device_id=...)`, device-resident records produced by the HIP
path, BOTH all-gathers of dist.gather_match_lists executed on the RCCL communicator (the world-size-1 early return
bypassed), pair order checked.  What an 8-GPU run adds to this is peers, not code."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from featurematching_amd import dist as fdist
from featurematching_amd import ops, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_match_records_from_the_hip_path_go_through_rccl_all_gathers():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT")}
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        # three pairs through the HIP coarse stage: their records are device tensors the library wrote
        hw = (16, 16)
        f0, f1 = synth.coarse_descriptors(61, 3, hw[0] * hw[1], 64, "peaky")
        out = ops.coarse_match(torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev), hw, hw, 8.0)
        m = out['i_ids'].shape[0]
        assert m > 100 and out['mconf'].is_cuda
        rec = fdist.pack_records(out['b_ids'], out['mkpts0_c'], out['mkpts1_c'], out['mconf'], pair_offset=40)
        assert rec.is_cuda and rec.shape == (m, fdist.RECORD)
        same = fdist.gather_match_lists(rec)                              # a world of one: nothing to exchange
        assert same is rec
        full = fdist.gather_match_lists(rec, always_exchange=True)        # counts + padded records over RCCL
        torch.cuda.synchronize()
        assert full.is_cuda and torch.equal(full, rec)
        ids, k0, k1, conf = fdist.unpack_records(full)
        assert bool((ids[1:] >= ids[:-1]).all()) and int(ids[0]) == 40 and int(ids[-1]) == 42
        assert torch.equal(k0, out['mkpts0_c']) and torch.equal(k1, out['mkpts1_c']) and torch.equal(conf, out['mconf'])
        # an empty list travels too (a rank whose pairs found nothing)
        empty = fdist.gather_match_lists(rec[:0], always_exchange=True)
        assert empty.shape == (0, fdist.RECORD)
        # the other collectives bench.py's N > 1 path issues
        t = torch.tensor([3.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        got = [None]
        dist.all_gather_object(got, "rank 0")
        assert float(t[0]) == 3.5 and got == ["rank 0"]
    finally:
        dist.destroy_process_group()
        for k, v in saved.items():             # (the rendezvous variables must not leak into later tests' child processes)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_bench_under_a_launcher_environment_prints_the_distributed_block():
    """`bench.py --gpus 1` with torchrun's environment (WORLD_SIZE = 1): the RCCL group is initialised with device_id, the
    match lists of the timed steps go through both all-gathers, and the JSON line names backend, world size and the
    device.  A child process (the test process keeps its own GPU state); --quick --skip-cpu: seconds."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "8",
                        "--quick", "--skip-cpu", "--pairs", "4"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    d = out["distributed"]
    assert d["backend"].startswith("nccl") and d["world_size"] == 1 and len(d["device_uuids"]) == 1
    assert out["n_gpus"] == 1 and out["gathered_records"] > 3000 and out["gather_ms"] > 0 and out["verified"]


def test_two_ranks_on_one_gpu_run_the_hip_step_and_gather_over_gloo():
    """`python bench.py --gpus 2` (the launcher form the driver uses for N > 1: the parent starts torch.distributed.run
    itself) with FM_BENCH_BACKEND=gloo: two ranks share the one GPU of the test box, each runs the real HIP step on its own
    block of pairs, the match lists of the timed steps are packed on the device, gathered over gloo in pair order, and rank
    0 prints ONE JSON line that says so.  Everything of the N > 1 path except RCCL between several devices."""
    env = dict(os.environ, FM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "4",
                        "--quick", "--skip-cpu", "--pairs", "4"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["verified"]
    d = out["distributed"]
    assert d["backend"].startswith("gloo") and d["world_size"] == 2 and len(d["device_uuids"]) == 2
    assert out["gathered_records"] > 2 * 4 * 3000          # both ranks' lists (4 input sets each, ~3.8 k matches per pair)
    assert out["config"]["pair_block"] == [0, 1]           # rank 0's block of the 2-pair global step
