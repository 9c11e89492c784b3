"""The step after the matching path: per-match epipolar errors, per-pair inlier scores (HIP kernel
`k_epipolar`, csrc/post.hip) and the wire / on-disk format of match lists.

`compute_symmetrical_epipolar_errors(data)` has the reference's name, reads the reference's keys
(utils/metrics.py:60-81: T_0to1 [N,4,4], K0/K1 [N,3,3], m_bids, mkpts0_f, mkpts1_f) and writes data['epi_errs'];
the reference loops over the batch on the host and builds E with kornia, here one launch does the whole list.

Wire format (little-endian): a 16-byte header {magic "FMT1", uint32 version = 1,
uint32 record_bytes = 24, uint32 count} followed by `count` records {int32 pair_id, float32 x0, y0, x1, y1, conf} -
the records `dist.pack_records` produces and `dist.gather_match_lists` exchanges between ranks.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Optional

import numpy as np
import torch

from . import _lib
from .dist import RECORD, pack_records, unpack_records

MAGIC = b"FMT1"
VERSION = 1


def epipolar_errors(mkpts0: torch.Tensor, mkpts1: torch.Tensor, m_bids: torch.Tensor, T_0to1: torch.Tensor,
                    K0: torch.Tensor, K1: torch.Tensor, inlier_thr: float = 1e-4, count: Optional[torch.Tensor] = None):
    """-> (epi_errs float32 [M], inlier bool [M], per_pair int32 [N,2] = matches / inliers per pair)."""
    lib = _lib.load()
    if not mkpts0.is_cuda:
        raise RuntimeError("mkpts0 must live on the GPU: the HIP path has no CPU fallback")
    dev = mkpts0.device
    k0, k1 = mkpts0.float().contiguous(), mkpts1.float().contiguous()
    m, stride = k0.shape
    n = T_0to1.shape[0]
    f = lambda t: t.to(dev).float().contiguous()
    T, a0, a1 = f(T_0to1), f(K0), f(K1)
    epi = torch.empty(m, dtype=torch.float32, device=dev)
    inl = torch.zeros(m, dtype=torch.uint8, device=dev)
    per = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    if m:
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        st = lib.fm_epipolar_errors(p(k0), p(k1), stride, p(m_bids.to(dev).long().contiguous()), p(count), m, n, p(T),
                                    p(a0), p(a1), float(inlier_thr), p(epi), p(inl), p(per),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(st, "fm_epipolar_errors")
    return epi, inl.bool(), per


def compute_symmetrical_epipolar_errors(data: dict) -> None:
    """Drop-in for utils/metrics.py:60-81: data['epi_errs'] [M] from data['T_0to1'], 'K0', 'K1', 'm_bids',
    'mkpts0_f', 'mkpts1_f'."""
    epi, _, _ = epipolar_errors(data['mkpts0_f'], data['mkpts1_f'], data['m_bids'], data['T_0to1'], data['K0'], data['K1'])
    data.update({'epi_errs': epi})


def dumps(records: torch.Tensor) -> bytes:
    """int32 [M,6] records (dist.pack_records) -> bytes in the wire format."""
    rec = records.detach().to("cpu", torch.int32).contiguous().numpy().astype("<i4", copy=False)
    if rec.ndim != 2 or rec.shape[1] != RECORD:
        raise ValueError(f"records must be [M, {RECORD}] int32")
    return MAGIC + struct.pack("<III", VERSION, RECORD * 4, rec.shape[0]) + rec.tobytes()


def loads(buf: bytes) -> torch.Tensor:
    if len(buf) < 16 or buf[:4] != MAGIC:
        raise ValueError("not a match-list stream (bad magic)")
    version, rbytes, count = struct.unpack("<III", buf[4:16])
    if version != VERSION or rbytes != RECORD * 4:
        raise ValueError(f"unsupported match-list stream: version {version}, {rbytes}-byte records")
    if len(buf) != 16 + count * rbytes:
        raise ValueError(f"truncated match-list stream: {len(buf)} bytes for {count} records")
    rec = np.frombuffer(buf, dtype="<i4", offset=16).reshape(count, RECORD)
    return torch.from_numpy(rec.astype(np.int32))


def save_matches(path: str, m_bids, kpts0, kpts1, conf, pair_offset: int = 0) -> int:
    rec = pack_records(m_bids, kpts0, kpts1, conf, pair_offset)
    with open(path, "wb") as f:
        f.write(dumps(rec))
    return int(rec.shape[0])


def load_matches(path: str):
    """-> (pair_ids int64 [M], kpts0 [M,2], kpts1 [M,2], conf [M])"""
    with open(path, "rb") as f:
        return unpack_records(loads(f.read()))
