"""Pair sharding + match-list gather on CPU with the gloo backend (world size 2 and 3)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from featurematching_amd import dist as fdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_matches(pair_lo, pair_hi, seed=0):
    """Deterministic 'match lists' of a block of pairs: M varies per pair (including 0)."""
    g = np.random.default_rng(seed)
    recs = []
    for p in range(pair_lo, pair_hi):
        rs = np.random.default_rng(1000 + p)
        m = int(rs.integers(0, 7)) if p % 4 else 0
        b = np.full(m, p - pair_lo, np.int64)
        k0 = rs.random((m, 2)).astype(np.float32) * 640
        k1 = rs.random((m, 2)).astype(np.float32) * 640
        c = rs.random(m).astype(np.float32)
        recs.append((b, k0, k1, c))
    cat = lambda i: torch.as_tensor(np.concatenate([r[i] for r in recs]) if recs else np.zeros((0,) + ((2,) if i in (1, 2) else ())))
    return cat(0).long(), cat(1).float().reshape(-1, 2), cat(2).float().reshape(-1, 2), cat(3).float()


def _worker(rank, world, port, num_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = fdist.shard_range(num_pairs, rank, world)
    b, k0, k1, c = _fake_matches(lo, hi)
    rec = fdist.pack_records(b, k0, k1, c, pair_offset=lo)
    full = fdist.gather_match_lists(rec)
    torch.save(full, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_pairs", [(2, 9), (3, 8), (2, 1)])
def test_gather_equals_single_process(tmp_path, world, num_pairs):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_pairs, str(tmp_path)), nprocs=world, join=True)
    b, k0, k1, c = _fake_matches(0, num_pairs)
    ref = fdist.pack_records(b, k0, k1, c, 0)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert torch.equal(got, ref), f"rank {r}"
    ids, g0, g1, gc = fdist.unpack_records(ref)
    assert torch.equal(ids, b) and torch.equal(gc, c)
    assert np.all(np.diff(ids.numpy()) >= 0)             # rank-major == pair order


def _recorded():
    """Match lists the HIP path produced on an MI355X for 4 pairs of config #1 (tools/record_hip_matches.py)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hip_matches_cfg1x4.npz"))
    return (torch.as_tensor(z["b_ids"]).long(), torch.as_tensor(z["kpts0"]), torch.as_tensor(z["kpts1"]),
            torch.as_tensor(z["mconf"]))


def _worker_recorded(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, k0, k1, c = _recorded()
    lo, hi = fdist.shard_range(4, rank, world)
    mine = (b >= lo) & (b < hi)                       # this rank's block of pairs, local batch ids from 0
    rec = fdist.pack_records(b[mine] - lo, k0[mine], k1[mine], c[mine], pair_offset=lo)
    torch.save(fdist.gather_match_lists(rec), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_of_recorded_hip_outputs(tmp_path):
    """cfg#4's exchange on real outputs: two ranks each hold the HIP match lists of their two pairs; after
    pack -> all-gather -> unpack every rank holds the single-process list, bit for bit and in pair order."""
    port = _free_port()
    mp.spawn(_worker_recorded, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b, k0, k1, c = _recorded()
    assert b.shape[0] > 100 and sorted(set(b.tolist())) == [0, 1, 2, 3]
    ref = fdist.pack_records(b, k0, k1, c, 0)
    assert ref.dtype == torch.int32 and ref.shape[1] * 4 == 24          # 24-byte records, int32 pair id
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert torch.equal(got, ref), f"rank {r}"
        ids, g0, g1, gc = fdist.unpack_records(got)
        assert torch.equal(ids, b) and torch.equal(g0, k0) and torch.equal(g1, k1) and torch.equal(gc, c)


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    b, k0, k1, c = _fake_matches(0, 5)
    rec = fdist.pack_records(b, k0, k1, c)
    assert fdist.gather_match_lists(rec) is rec


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks (torch.distributed.run) before
    any GPU call and passes rank 0's ONE JSON line through.  --stub-step replaces the HIP step with a CPU stand-in, so
    the launcher, the process group, the pair blocks, the barrier-bracketed timed region with its max over ranks and
    the match-list gather run here; the line is marked as a stub (no measurement)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-step", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    # stdout is ONE JSON line and nothing else (what the ranks' libraries print - gloo's connection notes - is on stderr)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout
    out = json.loads(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["stub"] is True and out["value"] is None
    assert out["gather_ms"] > 0
    # two ranks, pair blocks [0,1) and [1,2): 50 + 57 stand-in records, gathered on every rank in pair order
    assert out["gathered_records"] == 107 and out["config"]["pair_block"] == [0, 1]
