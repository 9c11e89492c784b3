"""Pair-parallel sharding across the GPUs of a node and the gather of match lists.

Every pair is independent in all three stages, so ranks take contiguous blocks of
pairs and nothing is exchanged on the data path.  The only collective is the gather of the
resulting match lists: one all-gather of the per-rank counts and one all-gather of records padded
to the largest count - the pad-to-largest scheme of the reference's utils/comm.py:113-176, but
with raw 24-byte records over RCCL (backend "nccl" on ROCm) instead of pickles over gloo.

Record (little-endian, 24 bytes) = {int32 pair_id, float32 x0, y0, x1, y1, conf}; a list of
records is an int32 tensor [M, 6] whose columns 1..5 hold the float bits.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

RECORD = 6


def shard_range(num_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of pairs owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(num_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(b_ids: torch.Tensor, kpts0: torch.Tensor, kpts1: torch.Tensor, conf: torch.Tensor,
                 pair_offset: int = 0) -> torch.Tensor:
    """[M, 6] int32 records {pair_id, bits(x0, y0, x1, y1, conf)}; local batch ids become global pair ids."""
    m = b_ids.shape[0]
    rec = torch.empty(m, RECORD, dtype=torch.int32, device=conf.device)
    rec[:, 0] = (b_ids + pair_offset).to(torch.int32)
    f = rec[:, 1:].view(torch.float32) if m else rec[:, 1:]
    if m:
        f[:, 0:2] = kpts0[:, :2].float()
        f[:, 2:4] = kpts1[:, :2].float()
        f[:, 4] = conf.float()
    return rec


def gather_match_lists(records: torch.Tensor, group=None, always_exchange: bool = False) -> torch.Tensor:
    """All ranks receive the concatenation (rank-major, each rank's order preserved) of every
    rank's records - identical to a single process run on the concatenated batch.

    Two collectives on the records' device (RCCL for device tensors, gloo for CPU ones): the per-rank counts into one
    [world] tensor - read back with ONE host copy, the sync the padding needs - and the records padded to the largest
    count into one [world * cap, 6] tensor (utils/comm.py:113-176's pad-to-largest scheme).  A world of one returns its
    records as they are unless `always_exchange` (the single-GPU rehearsal of the N-rank path: both collectives run)."""
    if not (dist.is_available() and dist.is_initialized()):
        return records
    world = dist.get_world_size(group)
    if world == 1 and not always_exchange:
        return records
    dev = records.device
    count = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts_t = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, count, group=group)
    counts = counts_t.tolist()                       # one device -> host copy for all ranks' counts
    cap = max(counts)
    padded = torch.zeros(cap, RECORD, dtype=torch.int32, device=dev)
    padded[:records.shape[0]] = records
    bufs = torch.empty(world * cap, RECORD, dtype=torch.int32, device=dev)       # rank r's block = rows [r cap, (r+1) cap)
    dist.all_gather_into_tensor(bufs, padded, group=group)
    return torch.cat([bufs[r * cap:r * cap + c] for r, c in enumerate(counts)], dim=0)


def unpack_records(rec: torch.Tensor):
    """-> (pair_ids int64 [M], kpts0 [M,2], kpts1 [M,2], conf [M])"""
    f = rec[:, 1:].contiguous().view(torch.float32)
    return rec[:, 0].to(torch.int64), f[:, 0:2], f[:, 2:4], f[:, 4]
