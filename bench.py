#!/usr/bin/env python3
"""Benchmark of the matching hot path (coarse correlation + dual-softmax mutual-NN assignment
-> window crop -> fine correlation + soft-argmax) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the path over one batch of synthetic feature maps that are already
resident in HBM.  Default workload = BASELINE.json configs[1]: one 640x480 pair, C=256 coarse
descriptors at 1/8 (L=S=4800), 64-d fine maps at 1/2, 5x5 fine window.  Every rank works on
its own block of pairs (dist.shard_range; weak scaling, no data-path collective); the timed region is
bracketed by a barrier + device synchronise and the maximum over ranks is reported.  With more than one
rank the match lists of the last step are gathered (dist.gather_match_lists: the one collective of the
path) and that exchange is timed separately ("gather_ms").  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
import platform
import sys
import time

# One hardware queue per HIP stream (+ the null stream): with the runtime's default of 4 queues two of the
# bench's streams share a queue and their kernels serialise (measured 9.6k -> 11.4k pairs/s at 640x480).
# Four pairs in flight measured best with inputs streamed from HBM (12 resident input sets); more than four
# active queues are time-sliced by the command processor (5+ streams lose 20 %).
# A runtime setting of the HIP process, read when the runtime initialises - hence before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops, synth  # noqa: E402
from featurematching_amd import dist as fdist  # noqa: E402

PEAK_F16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
PEAK_I8_DENSE_TOPS = 5000.0        # ... I8 MFMA: the cycles of the BF16 form at twice the k
HBM_PEAK_GBS = 8000.0              # ... HBM3E 8 TB/s spec (6.3 TB/s measured with a float4 copy)
WORKLOADS = {
    "cfg2": dict(n=1, h=480, w=640, c=256, cf=64, label="640x480 pair, C=256 @1/8 (L=S=4800), Cf=64 @1/2"),
    "cfg3": dict(n=64, h=480, w=640, c=256, cf=64, label="batch of 64 640x480 pairs, C=256 @1/8"),
    "cfg5": dict(n=1, h=1024, w=1024, c=256, cf=64, label="1024x1024 pair, C=256 @1/8 (L=S=16384)"),
    "cfg1": dict(n=1, h=128, w=128, c=64, cf=64, label="128x128 pair, C=64 @1/8 (L=S=256)"),
    "l9600": dict(n=1, h=640, w=960, c=256, cf=64, label="640x960 pair, C=256 @1/8 (L=S=9600: the 9600 x 9600 cost volume)"),
}


class Pair:
    """Device-resident inputs of one batch of pairs plus the launch of one step on them.  The window
    buffers belong to the stream the pair runs on (`share`: pairs of one stream run one after the other)."""

    def __init__(self, wl, seed, window, dev, dist, share=None, device_data=False, layout="nchw", fine_path="maps"):
        sh = synth.config_shapes(wl)
        self.seed, self.dist, self.wl = seed, dist, wl
        self.n, self.l, self.c = wl["n"], sh["l"], wl["c"]
        self.hw_c, self.hw_f, self.hw_i = (sh["hc"], sh["wc"]), (sh["hf"], sh["wf"]), (wl["h"], wl["w"])
        self.window = window
        self.device_data = device_data or self.n > 4
        if not self.device_data:
            f0, f1 = synth.coarse_descriptors(seed, self.n, self.l, self.c, dist)
            self.f0, self.f1 = torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev)
            ff0, ff1 = synth.fine_maps(seed, self.n, wl["cf"], sh["hf"], sh["wf"])
            self.ff0, self.ff1 = torch.as_tensor(ff0, device=dev), torch.as_tensor(ff1, device=dev)
        else:   # secondary lines / large batches: the same statistics generated on the device (the portable numpy
                # hash RNG takes ~2 s per 640x480 pair); verification reads the tensors back
            g = torch.Generator(device=dev).manual_seed(seed)
            gain, sigma = synth.DISTRIBUTIONS[dist]
            self.f0 = gain * torch.randn(self.n, self.l, self.c, device=dev, generator=g)
            self.f1 = torch.empty_like(self.f0)
            for b in range(self.n):
                perm = torch.randperm(self.l, device=dev, generator=g)
                self.f1[b] = self.f0[b][perm] + sigma * torch.randn(self.l, self.c, device=dev, generator=g)
            if dist == "mixed":
                self.f0[torch.rand(self.n, self.l, device=dev, generator=g) < synth.MIXED_FRACTION] *= synth.MIXED_SCALE
                self.f1[torch.rand(self.n, self.l, device=dev, generator=g) < synth.MIXED_FRACTION] *= synth.MIXED_SCALE
            self.ff0 = torch.randn(self.n, wl["cf"], sh["hf"], sh["wf"], device=dev, generator=g)
            self.ff1 = torch.randn(self.n, wl["cf"], sh["hf"], sh["wf"], device=dev, generator=g)
        self.layout, self.fine_path = layout, fine_path
        if layout == "nhwc":      # channels-last storage of the same logical [N,Cf,Hf,Wf] maps
            self.ff0 = self.ff0.contiguous(memory_format=torch.channels_last)
            self.ff1 = self.ff1.contiguous(memory_format=torch.channels_last)
        w0, b0, w1, b1 = synth.mix_weights(seed, window * window)
        self.mix = (w0, b0, w1, b1)
        self.mix0 = torch.as_tensor(np.concatenate([w0, [b0]]).astype(np.float32), device=dev)
        self.mix1 = torch.as_tensor(np.concatenate([w1, [b1]]).astype(np.float32), device=dev)
        self.cap = self.n * self.l
        self.win0 = self.win1 = self.scratch = None
        if share is not None:
            self.win0, self.win1, self.scratch = share.win0, share.win1, share.scratch
        elif fine_path == "windows":
            self.win0 = torch.empty(self.cap, window * window, wl["cf"], device=dev)
            self.win1 = torch.empty_like(self.win0)
        elif layout == "nchw":    # channels-last copy of image 1 (fm_fine_match_maps makes it per call)
            self.scratch = torch.empty(self.ff1.numel() * 4, dtype=torch.uint8, device=dev)
        self.last = None
        self.gather = "cells"    # cells | list (see ops.gather_windows)
        # Which launches the asynchronous coarse call enqueues is NOT hand-picked per distribution: the library's own
        # one-call entry point (fm_coarse_match_auto through ops.coarse_match) serves this input set twice - the first call
        # finds out what the data needs (flat similarity: the dense sum kernel; rows without a partner next to peaked ones:
        # 16 candidate slots + the exact int8 step), the second starts from the hint word the first one left and reports
        # whether EVERY sample went to the dense kernel (then FM_MODE_FLAT: no screening sweep) - and the step replays the
        # mode that hint names.  'peaky' data: hint 0, the common path's launches.
        self.hint = learnt_hint(self)
        d = ops.HintMemory.decode(self.hint)
        self.dense, self.exact, self.flat, self.exact_step = d['dense'], d['exact'], d['flat'], d['step']
        self.slots = d['slots'] or None                   # candidate slots per row / column (None: fm_default_cand_slots(thr))
        # NCHW float32 maps on the maps path: image 1's channels-last copy rides in the assignment kernel's launch
        # (fm_coarse_match_maps) instead of being fm_fine_match_maps' first launch
        self.fuse_maps = layout == "nchw" and fine_path == "maps"
        self.conf_matrix = False         # materialise data['conf_matrix'] (cfg#3's HBM-bound mode)
        self.alone = False               # FM_MODE_ALONE: only the ONE-stream line sets it (that step does have the GPU to itself)
        self.stages = "all"      # diagnostic only (--stages): "coarse" or "fine" time a part of the step

    def step(self):
        """Enqueue the whole path; nothing synchronises the host (the match count stays on the
        device and the window/fine kernels read it there)."""
        w = self.window
        if self.stages == "fine" and self.last is not None:
            buf = self.last[0]
        else:
            buf = ops.coarse_match_async(self.f0, self.f1, self.hw_c, self.hw_c, self.hw_i[0] / self.hw_c[0],
                                         cap=self.cap, cand_slots=self.slots, dense=self.dense, exact_screening=self.exact,
                                         exact_step=self.exact_step,
                                         conf_matrix=self.conf_matrix, flat=(self.flat and self.dense and not self.conf_matrix),
                                         side_map=(self.ff1 if self.fuse_maps else None),
                                         side_scratch=(self.scratch if self.fuse_maps else None),
                                         cell_maps=(self.fine_path == "windows" and self.layout == "nchw"), alone=self.alone)
        if self.stages == "coarse" and self.last is not None:
            self.last = (buf,) + self.last[1:]
            return self.last
        if self.fine_path == "maps":
            k0, k1 = self.fine_maps(buf)
        else:
            self.crop(buf)
            k0, k1 = self.fine(buf)
        self.last = (buf, k0, k1)
        return self.last

    def fine_maps(self, buf, standalone=False):
        """window crop + fine stage from the maps in one call (no window tensors); standalone: with its own transpose of
        image 1 even when the step lets the coarse call carry it"""
        prepared = self.scratch if (self.fuse_maps and not standalone) else None
        return ops.fine_match_maps(self.ff0, self.ff1, buf.b_ids, buf.i_ids, buf.j_ids, self.window, 4, self.hw_c[1],
                                   self.hw_c[1], self.mix0, self.mix1, buf.mkpts0_c, buf.mkpts1_c,
                                   self.hw_i[0] / self.hw_f[0], count=buf.count, scratch=self.scratch, prepared=prepared)

    def crop(self, buf):
        w = self.window
        if self.layout == "nhwc":       # channels-last maps: one list-ordered copy kernel per image
            ops.gather_windows(self.ff0, buf.b_ids, buf.i_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win0)
            ops.gather_windows(self.ff1, buf.b_ids, buf.j_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win1)
        elif self.gather == "cells":    # both images' crops in one launch, cell order
            ops.gather_windows_pair(self.ff0, self.ff1, buf.b_ids, buf.i_ids, buf.j_ids, w, 4, self.hw_c, self.hw_c,
                                    buf.cell_maps(), count=buf.count, out0=self.win0, out1=self.win1)
        else:                           # "list": one list-ordered launch per image
            ops.gather_windows(self.ff0, buf.b_ids, buf.i_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win0)
            ops.gather_windows(self.ff1, buf.b_ids, buf.j_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win1)

    def fine(self, buf):
        return ops.fine_match(self.win0, self.win1, self.mix0, self.mix1, buf.mkpts0_c, buf.mkpts1_c,
                              self.hw_i[0] / self.hw_f[0], count=buf.count)


_HINTS = {}      # (workload shape, distribution) -> hint word of fm_coarse_match_auto


def learnt_hint(pair):
    """The hint word fm_coarse_match_auto leaves for this kind of data (two calls: learn, then confirm from the hint);
    learnt once per (shape, distribution) and process."""
    key = (pair.n, pair.l, pair.c, pair.dist)
    if key not in _HINTS:
        mem = ops.MODE_MEMORY.snapshot()
        ops.MODE_MEMORY.clear()
        h = 0
        for _ in range(3):
            out = ops.coarse_match(pair.f0, pair.f1, pair.hw_c, pair.hw_c, pair.hw_i[0] / pair.hw_c[0])
            h = out['_coarse_buffers'].hint
            del out
        ops.MODE_MEMORY.clear()
        for k, v in mem.items():
            ops.MODE_MEMORY.finish(k, v['hint'])
        torch.cuda.synchronize()
        _HINTS[key] = h
    return _HINTS[key]


def _events(fn, iters=10, before=None, group=6):
    """Average duration (ms) of ONE of fn()'s launches.  `group` back-to-back launches are captured into a hipGraph
    (the Python / ctypes cost of one call, 5-30 us, exceeds these kernels' durations: enqueued eagerly the GPU
    would wait for the host) and every replay is bracketed by events on the stream it runs on; `before` (counter
    reset) is replayed outside the bracket.  All replays are enqueued before the host waits.  What remains in the
    figure is the ~1.5 us in-stream gap between dependent kernels, so the committed rocprofv3 trace (profiles/)
    reads that much lower."""
    st = torch.cuda.current_stream()
    g, gb = torch.cuda.CUDAGraph(), None
    with torch.cuda.graph(g, stream=st):
        for _ in range(group):
            fn()
    if before is not None:
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=st):
            before()
    evs = []
    for _ in range(iters + 2):
        if gb is not None:
            gb.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs[2:]) / (iters * group)


def time_kernels(pair):
    """Event-timed launches of the kernels of one step on a workspace the step has filled: the whole coarse stage (one
    fm_coarse_match call: all its launches with their in-stream gaps), its kernels k_prep_split, k_max_i8 and
    k_thresh + k_screen_rows alone (+ the float16 planes and the dense sum kernel when the pair runs with FM_MODE_DENSE), the
    window crop and the fine kernel.  The assignment kernel cannot be re-run on its own outputs; its share is what
    remains of the coarse stage."""
    lib = _lib.load()
    buf = pair.last[0]
    ws = buf.workspace
    ptr = C.c_void_p(ws.data_ptr() + ((-ws.data_ptr()) % 256))
    slots = pair.slots or lib.fm_default_cand_slots(0.2)
    f0, f1 = C.c_void_p(pair.f0.data_ptr()), C.c_void_p(pair.f1.data_ptr())
    shape = (pair.n, pair.l, pair.l, pair.c, slots)

    def st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def reset():
        _lib.check(lib.fm_debug_reset_counters(ptr, *shape, st()), "reset")

    def sparse():
        _lib.check(lib.fm_debug_launch_screen(ptr, f0, f1, *shape, 0.1, 0.2, st()), "sparse")

    t = {}
    t["max"] = _events(lambda: _lib.check(lib.fm_debug_launch_corr(ptr, *shape, 0.1, 0.2, 0, st()), "max"))
    # (4 launches per reset: a row's 8 candidate slots take the one candidate each launch adds on 'peaky' data)
    flat = pair.dense and pair.flat and not pair.conf_matrix
    t["planes"] = t["dense"] = 0.0
    if flat:
        # FM_MODE_FLAT: no screening sweep and no plane kernel - the stabiliser kernel k_stab takes the sweep's place, the
        # planes come out of k_prep_split (timed as "prep" below); the dense sum kernel redoes every sample
        def stab():
            _lib.check(lib.fm_debug_launch_flat(ptr, f0, f1, *shape, 0.1, 0.2, 1, st()), "stab")

        def reflag():          # (the counters the dense kernel appends to are cleared, k_stab flags the samples again)
            reset()
            stab()
        t["sparse"] = _events(stab)
        t["dense"] = _events(lambda: _lib.check(lib.fm_debug_launch_corr(ptr, *shape, 0.1, 0.2, 1, st()), "dense"),
                             before=reflag, group=1, iters=20)
    else:
        t["sparse"] = _events(sparse, before=reset, group=4 if pair.dist == "peaky" else 1, iters=10 if pair.dist == "peaky" else 20)
    if pair.dense and not flat:
        # the dense sum kernel redoes the samples the sparse one flagged; the untimed part of every iteration clears
        # the counters and lets the sparse kernel flag again
        def reflag():
            reset()
            sparse()
        reflag()
        t["planes"] = _events(lambda: _lib.check(lib.fm_debug_launch_prep_f16(ptr, f0, f1, *shape, 0, st()), "planes"))
        # every launch redoes the flagged samples and adds their candidates: one launch per reflag
        t["dense"] = _events(lambda: _lib.check(lib.fm_debug_launch_corr(ptr, *shape, 0.1, 0.2, 1, st()), "dense"),
                             before=reflag, group=1, iters=20)
    if flat:
        t["prep"] = _events(lambda: _lib.check(lib.fm_debug_launch_flat(ptr, f0, f1, *shape, 0.1, 0.2, 0, st()), "prep+planes"))
    else:
        t["prep"] = _events(lambda: _lib.check(lib.fm_debug_launch_prep(ptr, f0, f1, *shape, st()), "prep"))
    # the whole coarse stage, as the step enqueues it
    keep = pair.stages
    pair.stages = "coarse"
    t["coarse"] = _events(pair.step, group=4)
    # restore a consistent workspace for the crop / fine timings below
    pair.stages = "all"
    pair.step()
    torch.cuda.synchronize()
    buf = pair.last[0]
    if pair.fine_path == "maps":
        t["crop"] = 0.0
        # (NCHW: image 1's transpose + the fused kernel, as two launches of its own - also when the step lets the coarse
        # call's assignment launch carry the transpose: then "fine_prepared" is what follows the coarse stage)
        t["fine"] = _events(lambda: pair.fine_maps(buf, standalone=True), group=3)
        t["fine_prepared"] = _events(lambda: pair.fine_maps(buf), group=3) if pair.fuse_maps else t["fine"]
    else:
        t["crop"] = _events(lambda: pair.crop(buf), group=3 if pair.layout == "nhwc" else 6)
        t["fine"] = _events(lambda: pair.fine(buf))
    pair.stages = keep
    return t


def copy_rate(nbytes, dev, iters=24):
    """GB/s (read + write) of a plain device-to-device copy that moves `nbytes` in total: the practical ceiling of an
    HBM-bound stream of that size on this GPU, next to the 8 TB/s datasheet figure the fractions are quoted against."""
    n = max(1, int(nbytes) // 8)                       # float32 elements read (and as many written)
    srcs = [torch.empty(n, device=dev).normal_() for _ in range(6)]      # 6 x ~48 MB: no single buffer stays hot
    dst = torch.empty(n, device=dev)
    for sbuf in srcs[:2]:
        dst.copy_(sbuf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        dst.copy_(srcs[i % len(srcs)])
    e1.record()
    torch.cuda.synchronize()
    return 8.0 * n / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9


def kernel_source_sha():
    """sha256 (first 16 hex digits) over the HIP sources of the library: the committed counter files carry the value
    they were collected with, and a file whose value differs from the sources in this tree is stale."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "featurematching_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def committed_traffic(workload):
    """Fabric/HBM bytes per launch of the correlation kernels from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_fetch_write_cfg2.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of `bench.py --streams 1
    --pairs 1 --no-graph`), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of the bytes of
    wide reads; both counters in KiB).  Counters collected by an earlier run, not by this one - hence the file name
    next to them.  The newest round's file whose `kernel_src_sha16` equals the sources in this tree is used (the
    counters then belong to these kernels); None when there is none or the workload is another one."""
    import glob
    if workload not in ("cfg2", "cfg3"):
        return None, None
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_fetch_write_{workload}.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_src_sha16") != sha:
            continue
        tot = 0.0
        for name, c in d.get("kernels", {}).items():
            if ("k_max_i8" in name or "k_screen" in name or "k_thresh" in name) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                tot += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if tot:
            return int(tot), os.path.relpath(path, ROOT)
    return None, None


def verify(pair, nverify=2):
    """Compare what the timed path produced for one input set with the CPU oracle (same tolerances as the parity
    tests): identical (b,i,j) outside the guard band |conf - thr| < 2e-5, mconf within 1e-5, fine keypoints
    within 1e-3 px.  Batches are checked on their first `nverify` samples (the oracle takes ~0.3 s per 640x480 pair)."""
    from oracle import matcher_ref as orc     # checker only
    nv = min(pair.n, nverify)
    f0, f1 = pair.f0[:nv].float().cpu().numpy(), pair.f1[:nv].float().cpu().numpy()
    ff0, ff1 = pair.ff0[:nv].cpu().numpy(), pair.ff1[:nv].cpu().numpy()
    ref = orc.match_features(f0, f1, ff0, ff1, pair.hw_i, pair.mix, w=pair.window)
    buf, k0, k1 = pair.last
    m = buf.read_count()
    got = {k: v.cpu().numpy() for k, v in buf.sliced(m).items()}
    sel = got["b_ids"] < nv
    k0, k1 = k0.cpu().numpy()[:m][sel], k1.cpu().numpy()[:m][sel]
    got = {k: v[sel] for k, v in got.items()}
    gk = {(int(b), int(i), int(j)): n for n, (b, i, j) in enumerate(zip(got["b_ids"], got["i_ids"], got["j_ids"]))}
    rk = {(int(b), int(i), int(j)): n for n, (b, i, j) in enumerate(zip(ref["b_ids"].numpy(), ref["i_ids"].numpy(),
                                                                          ref["j_ids"].numpy()))}
    rconf = ref["mconf"].numpy()
    stray = [k for k in gk if k not in rk and abs(got["mconf"][gk[k]] - 0.2) > 2e-5] + \
            [k for k in rk if k not in gk and abs(rconf[rk[k]] - 0.2) > 2e-5]
    common = [k for k in gk if k in rk]
    gi = np.array([gk[k] for k in common], dtype=np.int64)
    ri = np.array([rk[k] for k in common], dtype=np.int64)
    if not len(common):
        return {"ok": False, "matches": int(sel.sum()), "oracle_matches": len(rk)}
    conf_err = float(np.abs(got["mconf"][gi] - rconf[ri]).max())
    fine_err = max(float(np.abs(k0[gi] - ref["mkpts0_f"].numpy()[ri]).max()),
                   float(np.abs(k1[gi] - ref["mkpts1_f"].numpy()[ri]).max()))
    ok = not stray and conf_err <= 1e-5 and fine_err <= 1e-3 and abs(len(gk) - len(rk)) <= 4
    # how much of a check mconf is: on 'peaky' data every conf is 1.0 to the last bit
    nontrivial = int((rconf[ri] < 0.999).sum())
    return {"ok": bool(ok), "samples_checked": nv, "matches": len(gk), "oracle_matches": len(rk), "mconf_err": conf_err,
            "mconf_below_0.999": nontrivial, "fine_err_px": fine_err}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(wl, window, seed, budget_s=18.0):
    """The CPU oracle (a port of the reference's torch ops, pinned to the reference by the golden fixtures) on
    this host's cores, on a bounded sample of the same workload.  Rows: the matched-cells crop (what the HIP path
    computes) with 1 thread and with every thread count tried; the reference-shaped route (full F.unfold of every
    coarse cell, fine_preprocess.py:43-46, then select) and the reference's own window W = 7 beside the metric's
    W = 5.  `value` is the fastest row of the metric's configuration (W = `window`) over BOTH crop routes; torch's
    default of all hardware threads is several times slower for these memory-bound dense passes than a few cores."""
    from oracle import matcher_ref as orc     # cpu_baseline leg only
    sh = synth.config_shapes(wl)
    n = min(wl["n"], 1)
    f0, f1 = synth.coarse_descriptors(seed, n, sh["l"], wl["c"], "peaky")
    ff0, ff1 = synth.fine_maps(seed, n, wl["cf"], sh["hf"], sh["wf"])
    cores = os.cpu_count() or 8

    def rate(threads, w, unfold, share):
        torch.set_num_threads(threads)
        mix = synth.mix_weights(seed, w * w)
        run = lambda: orc.match_features(f0, f1, ff0, ff1, (wl["h"], wl["w"]), mix, w=w, use_unfold=unfold)
        run()                                                    # warm-up
        t0 = time.perf_counter()
        done = 0
        while done < 2 or (time.perf_counter() - t0 < budget_s * share and done < 200):
            run()
            done += n
        return round(done / (time.perf_counter() - t0), 3), done

    counts = sorted({c for c in (1, 8, 16, 32, 64) if c <= cores} | {min(8, cores)})
    rows = []
    for c in counts:
        v, d = rate(c, window, False, 0.3 if c in (1, 8) else 0.06)
        rows.append({"threads": c, "window": window, "crop": "matched cells", "value": v, "pairs_timed": d})
    best = max(rows, key=lambda r: r["value"])
    tb = best["threads"]
    v, d = rate(tb, window, True, 0.12)
    rows.append({"threads": tb, "window": window, "crop": "F.unfold of every cell (reference-shaped)", "value": v, "pairs_timed": d})
    if window != 7:
        v, d = rate(tb, 7, False, 0.08)
        rows.append({"threads": tb, "window": 7, "crop": "matched cells", "value": v, "pairs_timed": d})
        v, d = rate(tb, 7, True, 0.08)
        rows.append({"threads": tb, "window": 7, "crop": "F.unfold of every cell (reference-shaped)", "value": v, "pairs_timed": d})
    torch.set_num_threads(min(8, cores))
    single = next(r["value"] for r in rows if r["threads"] == 1 and r["crop"] == "matched cells" and r["window"] == window)
    # the stated baseline = the FASTEST row of the metric's window over both crop routes (the reference's own F.unfold
    # route was the faster one on the round-5 host: 7.34 against 6.28 pairs/s)
    top = max((r for r in rows if r["window"] == window), key=lambda r: r["value"])
    return {"value": top["value"], "unit": "image-pairs/s", "cores": top["threads"], "kind": "port",
            "route": top["crop"],
            "host_cores": cores, "cpu_model": cpu_model(), "single_thread_value": single,
            "thread_counts_tried": counts, "rows": rows,
            "sample": f"{top['pairs_timed']} x ({wl['label']}, {window}x{window} window) with {top['threads']} threads: "
                      f"oracle.match_features (torch-CPU ops mirroring the reference), crop route '{top['crop']}' - the "
                      f"fastest row of this window size; the other rows (thread counts {counts}, the other crop route, "
                      f"W = 7) are bounded samples of the same pair"}


def stream_rate(wl, window, dev, dist, batch, nstreams, steps=240, nsets=6, check=True, layout="nchw", fine_path="maps",
                flat_hint=True, slots=None, exact=None, exact_step=False, alone=False):
    """Pairs/s of the same step for another workload / distribution / pairs per launch: `nsets` resident input sets
    (generated on the device) cycled through on `nstreams` streams by hipGraph replay, `steps` timed steps.  Returns
    (pairs/s, verification of the first input set's last step against the oracle, matches per pair)."""
    wb = dict(wl, n=batch)
    pairs = []
    for p in range(nsets):
        pairs.append(Pair(wb, 5000 + 31 * p, window, dev, dist, share=pairs[p % nstreams] if p >= nstreams else None,
                          device_data=True, layout=layout, fine_path=fine_path))
        pairs[-1].flat = pairs[-1].flat and flat_hint
        pairs[-1].alone = alone
        if slots is not None:
            pairs[-1].slots = slots
        pairs[-1].exact_step = pairs[-1].exact_step or exact_step
        if exact is not None:
            pairs[-1].exact = exact
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for i, p in enumerate(pairs):
        with torch.cuda.stream(streams[i % nstreams]):
            p.step()
    torch.cuda.synchronize()
    graphs = []
    for i, p in enumerate(pairs):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i % nstreams]):
            p.step()
        graphs.append(g)

    def run(i):
        with torch.cuda.stream(streams[(i % nsets) % nstreams]):
            graphs[i % nsets].replay()

    # warm-up until the clocks have settled (these lines follow CPU-side verification pauses: the first tens of
    # milliseconds after one run at idle clocks - the 'borderline' line read 10.9 k against 13.8 k for the same step in
    # the headline's own timed region), then the median of three timed regions of `steps` steps
    tw = time.perf_counter()
    i = 0
    while i < 2 * nsets or time.perf_counter() - tw < 0.4:
        run(i)
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            run(i)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    with torch.cuda.stream(streams[0]):
        ms = [p.last[0].read_count() for p in pairs]
        assert min(ms) > 0
        ver = verify(pairs[0]) if check else None
    return batch * steps / dt, ver, float(np.mean(ms)) / batch


def module_api_rate(wl, window, dev, iters=60):
    """Pairs/s through the drop-in modules (modules.CoarseMatching -> FinePreprocess(no context merge) ->
    FineMatching): the reference-shaped call with its host sync (read_count) and per-call allocations."""
    from featurematching_amd import modules
    sh = synth.config_shapes(wl)
    f0, f1 = synth.coarse_descriptors(4242, wl["n"], sh["l"], wl["c"], "peaky")
    ff0, ff1 = synth.fine_maps(4242, wl["n"], wl["cf"], sh["hf"], sh["wf"])
    f0, f1, ff0, ff1 = (torch.as_tensor(x, device=dev) for x in (f0, f1, ff0, ff1))
    cm = modules.CoarseMatching({'thr': 0.2, 'border_rm': 2, 'dsmax_temperature': 0.1}).eval()
    fm = modules.FineMatching(window=window).to(dev).eval()
    hw_c, hw_f = (sh["hc"], sh["wc"]), (sh["hf"], sh["wf"])

    def once():
        data = {'hw0_i': (wl["h"], wl["w"]), 'hw1_i': (wl["h"], wl["w"]), 'hw0_c': hw_c, 'hw1_c': hw_c,
                'hw0_f': hw_f, 'hw1_f': hw_f, 'bs': wl["n"]}
        cm(f0, f1, data)
        buf = data['_fm_coarse']
        w0, w1 = ops.gather_windows_pair(ff0, ff1, data['b_ids'], data['i_ids'], data['j_ids'], window, 4, hw_c, hw_c,
                                         buf.cell_maps())
        fm(w0, w1, data)
        return data

    for _ in range(5):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    return wl["n"] * iters / (time.perf_counter() - t0)


def context_layer_times(wl, dev, iters=10):
    """The layers either side of the hot path, HIP kernels against the PyTorch-ROCm modules, and
    the whole `net.forward` tail (network/net.py:66-83) through matcher.Matcher.forward_features, W = 7."""
    from featurematching_amd.matcher import Matcher
    sh = synth.config_shapes(wl)
    n, l, hc, wc = wl["n"], sh["l"], sh["hc"], sh["wc"]
    torch.manual_seed(0)
    m = Matcher().to(dev).eval()
    f0, f1 = synth.coarse_descriptors(4243, n, l, wl["c"], "peaky")
    ff0, ff1 = synth.fine_maps(4243, n, wl["cf"], sh["hf"], sh["wf"])
    x0, x1, ff0, ff1 = (torch.as_tensor(x, device=dev) for x in (f0, f1, ff0, ff1))
    fc0 = x0.view(n, hc, wc, -1).permute(0, 3, 1, 2).contiguous()
    fc1 = x1.view(n, hc, wc, -1).permute(0, 3, 1, 2).contiguous()
    base = {'bs': n, 'hw0_i': (wl["h"], wl["w"]), 'hw1_i': (wl["h"], wl["w"])}

    def timed(fn):
        with torch.no_grad():
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, out

    res = {}
    t_c, _ = timed(lambda: m.coarse(x0, x1))
    d = dict(base, hw0_c=(hc, wc), hw1_c=(hc, wc), hw0_f=(sh["hf"], sh["wf"]), hw1_f=(sh["hf"], sh["wf"]))
    with torch.no_grad():
        m.coarse_matching(x0, x1, d)
        w0, w1 = m.fine_preprocess(ff0, ff1, x0, x1, d)
    t_f, _ = timed(lambda: m.fine(w0, w1))
    t_all, _ = timed(lambda: m.forward_features(fc0, fc1, ff0, ff1, dict(base)))
    m.coarse.use_hip = m.fine.use_hip = False           # the same modules through their torch ops
    try:
        t_c_t, _ = timed(lambda: m.coarse(x0, x1))
        t_f_t, _ = timed(lambda: m.fine(w0, w1))
    finally:
        m.coarse.use_hip = m.fine.use_hip = True
    nl = len(m.coarse.layer_names)
    flop_c = 2.0 * n * 2 * l * 655360 * nl
    mm = int(w0.shape[0])
    res["coarse"] = {"kernel": "k_ctx_kv + k_ctx_kv_sum + k_ctx_layer (hi/lo-split f16 MFMA: 3 products per float32 one)",
                     "layers": nl, "ms": round(t_c, 4), "torch_module_ms": round(t_c_t, 4),
                     "tflops_f32_equivalent": round(flop_c / t_c / 1e9, 1),
                     "frac_of_f16_mfma_peak": round(3.0 * flop_c / t_c / 1e9 / PEAK_F16_DENSE_TFLOPS, 3),
                     "note": "bound by the weight fragments per 32-token tile on the L2 -> CU path, and at one pair by the "
                             "tile count (150 tiles per image for 256 CUs)"}
    res["fine"] = {"kernel": "k_fine_tf<49> (hi/lo-split f16 MFMA, 32-token slices)", "matches": mm, "ms": round(t_f, 4),
                   "torch_module_ms": round(t_f_t, 4)}
    res["fine"]["range_fallbacks"] = int(m.fine.range_fallbacks)
    res["fine"]["start_log2_scale"] = int(getattr(m.fine, "_fine_start", 8))
    res["fine"]["note"] = ("the module starts a call at the activation scale the previous one ended at "
                           "(fm_fine_transformer_start): this random-weight network's activations leave 2^8 in nearly every "
                           "workgroup, and the first call's repeated passes (0.73 ms) are gone from the steady state")
    res["forward_features"] = {"ms": round(t_all, 4), "image_pairs_per_s": round(1e3 * n / t_all, 1),
                               "range_fallbacks": int(m.fine.range_fallbacks),
                               "note": "net.forward after the backbone: coarse context layers -> coarse matching -> "
                                       "crop + context merge -> fine context layers -> fine matching, eager, one pair "
                                       "per call, host sync on the match count"}
    return res


def masked_streams(dev, n, how, ncu=256):
    """Experiment (--cu-mask): n HIP streams that each own ncu / n compute units (hipExtStreamCreateWithCUMask), wrapped
    as torch external streams."""
    hip = C.CDLL("libamdhip64.so")
    out = []
    for k in range(n):
        bits = [(1 if ((i * n // ncu == k) if how == "blocks" else (i % n == k)) else 0) for i in range(ncu)]
        words = (C.c_uint32 * (ncu // 32))(*[sum(bits[32 * w + j] << j for j in range(32)) for w in range(ncu // 32)])
        st = C.c_void_p()
        err = hip.hipExtStreamCreateWithCUMask(C.byref(st), ncu // 32, words)
        if err != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask: {err}")
        out.append(torch.cuda.ExternalStream(st.value, device=dev))
    return out


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one process per
    GPU under torch.distributed.run (the launcher of the reference's training glue, utils/comm.py:113-176 assumes the
    same RANK / WORLD_SIZE environment), and pass rank 0's JSON line through.  Called before this process has made any
    GPU call: nothing that has initialised the GPU is ever replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # stdout carries ONE JSON line: whatever else the ranks' libraries write there (gloo's connection notes, ...) goes
    # to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        if line.startswith("{") and line.rstrip().endswith("}"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def under_launcher():
    """True when a torch.distributed launcher set this process up (RANK / WORLD_SIZE / MASTER_* in the environment) -
    also with WORLD_SIZE = 1: the one-GPU rehearsal of the N-rank path (`torchrun --nproc-per-node 1 bench.py --gpus 1`
    initialises the RCCL group with device_id, runs both all-gathers of the match-list exchange and prints the
    `distributed` block)."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init_ranks(world, backend, dev=None, grouped=False):
    """Process group of the bench's ranks: RCCL ("nccl") on the GPUs, "gloo" to rehearse the protocol on CPUs."""
    if world <= 1 and not grouped:
        return
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)


def timed_region(run, steps, reps, sync, world, coll_dev):
    """The contract's timed region, `reps` times: barrier + device synchronise, K steps, barrier + device synchronise;
    every repetition's duration is the MAXIMUM over the ranks.  Returns (durations, host enqueue times)."""
    import torch.distributed as dist

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    dts, enq = [], []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            run(i)
        enq.append(time.perf_counter() - t0)      # host time to enqueue the K steps (diagnostic: host- or GPU-bound?)
        barrier()
        dts.append(time.perf_counter() - t0)
    if world > 1:      # every repetition: the slowest rank
        td = torch.tensor(dts, device=coll_dev, dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in td.tolist()]
    return dts, enq


def agree_max(value, world, coll_dev):
    """the same integer on every rank (the largest any rank proposes)"""
    if world <= 1:
        return int(value)
    import torch.distributed as dist
    t = torch.tensor([int(value)], device=coll_dev, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def timed_gather(rec, world, sync, coll_dev, iters=10):
    """cfg#4's exchange: this rank's records (pair order) -> every rank holds all of them (dist.gather_match_lists:
    all-gather of counts + padded records).  Returns (ms per exchange, max over ranks; records gathered)."""
    import torch.distributed as dist
    full = fdist.gather_match_lists(rec, always_exchange=True)      # warm-up (communicator set-up)
    sync()
    dist.barrier()
    tg = time.perf_counter()
    for _ in range(iters):
        full = fdist.gather_match_lists(rec, always_exchange=True)
    sync()
    gather_ms = (time.perf_counter() - tg) / iters * 1e3
    ids = fdist.unpack_records(full)[0]
    assert bool((ids[1:] >= ids[:-1]).all()), "gathered records are not in pair order"
    t = torch.tensor([gather_ms], device=coll_dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0].item()), int(full.shape[0])


def main_stub(a, world, rank):
    """--stub-step: the launcher / rank protocol of this file (self-launch, process group, pair blocks, timed region
    with barriers and the max over ranks, match-list gather, rank 0's JSON line) with a CPU stand-in for the HIP step -
    what tests/test_dist.py drives on a machine without a GPU.  Its `value` is NOT a measurement of anything."""
    cpu = torch.device("cpu")
    init_ranks(world, "gloo")
    n = 1
    lo, hi = fdist.shard_range(world * n, rank, world)

    def fake(i):           # a deterministic 'match list' of this rank's pair block
        g = torch.Generator().manual_seed(1234 + lo)
        m = 50 + 7 * lo
        return (torch.zeros(m, dtype=torch.int64), torch.rand(m, 2, generator=g) * 640, torch.rand(m, 2, generator=g) * 640,
                torch.rand(m, generator=g))

    last = [None]

    def run(i):
        last[0] = fake(i)

    reps = agree_max(3, world, cpu)
    dts, _ = timed_region(run, a.steps, reps, lambda: None, world, cpu)
    dt = float(np.median(dts))
    out = {"metric": "STUB (no GPU work): rank protocol of bench.py only", "stub": True, "value": None, "unit": "image-pairs/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 6),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
           "config": {"workload": "stub step", "pair_block": [lo, hi]}}
    if world > 1:
        import torch.distributed as dist
        b, k0, k1, c = last[0]
        rec = fdist.pack_records(b, k0, k1, c, pair_offset=lo)
        out["gather_ms"], out["gathered_records"] = timed_gather(rec, world, lambda: None, cpu, iters=3)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--window", type=int, default=5, choices=[5, 7])
    ap.add_argument("--dist", default="peaky", choices=sorted(synth.DISTRIBUTIONS))
    ap.add_argument("--pairs", type=int, default=12,
                    help="distinct resident input sets cycled through (12 x 59 MB of inputs: far beyond the 256 MB "
                         "Infinity Cache, so every step reads its inputs from HBM)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the independent steps are spread over")
    ap.add_argument("--layout", default="nchw", choices=["nchw", "nhwc"],
                    help="storage of the fine maps: nchw = the reference's contiguous [N,Cf,Hf,Wf]; nhwc = channels-last")
    ap.add_argument("--fine-path", default="maps", choices=["windows", "maps"],
                    help="windows = window crop -> fine kernel; maps = crop + fine from the maps in one call (fm_fine_match_maps)")
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the timed K-step region (0 = enough for >= 120 ms of GPU work, at least 3)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--quick", action="store_true", help="skip the secondary lines (other workloads and data, module API)")
    ap.add_argument("--batch", type=int, default=0,
                    help="diagnostic: pairs per launch (overrides the workload's batch; the JSON line is then not the metric's config)")
    ap.add_argument("--stages", default="all", choices=["all", "coarse", "fine"],
                    help="diagnostic: time only a part of the step (the JSON line is then not the metric)")
    ap.add_argument("--no-fuse-maps", action="store_true",
                    help="A/B: fm_fine_match_maps transposes image 1 itself (its own launch) instead of the coarse call's "
                         "assignment launch carrying the copy (fm_coarse_match_maps)")
    ap.add_argument("--cu-mask", default="", choices=["", "blocks", "interleaved"],
                    help="experiment: every stream gets its own quarter (1 / streams) of the compute units "
                         "(hipExtStreamCreateWithCUMask): contiguous blocks of the mask bits, or interleaved bits")
    ap.add_argument("--alone", action="store_true",
                    help="diagnostic: every step passes FM_MODE_ALONE (grids sized for a kernel alone on the device); the JSON "
                         "line then says so in config.launch")
    ap.add_argument("--stub-step", action="store_true",
                    help="test hook: run the launcher / rank protocol with a CPU stand-in for the HIP step (no GPU needed; "
                         "the JSON line is marked as a stub and carries no measurement)")
    a = ap.parse_args()

    # `python bench.py --gpus N` (N > 1) without a launcher: this process becomes the launcher of N ranks.  Decided
    # before any GPU call of this process (torch is imported, nothing is initialised).
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.stub_step:
        return main_stub(a, world, rank)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FM_BENCH_BACKEND", "nccl")     # "gloo" to rehearse N ranks on one GPU
    if world > 1 and backend == "nccl" and ndev < world:
        sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, {ndev} visible (FM_BENCH_BACKEND=gloo rehearses "
                 f"the ranks on fewer)")
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    grouped = world > 1 or under_launcher()
    if grouped:
        import torch.distributed as dist
    init_ranks(world, backend, dev, grouped)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    wl = dict(WORKLOADS[a.workload])
    if a.batch > 0:
        wl["n"] = a.batch
        wl["label"] = f"batch of {a.batch}: " + wl["label"]
    # this rank's block of the global batch (world x n pairs per step)
    pair_lo, pair_hi = fdist.shard_range(world * wl["n"], rank, world)
    # enough distinct input sets to exceed the Infinity Cache several times over, not more (generating them
    # with the portable hash RNG is the slow part of the set-up)
    sh0 = synth.config_shapes(wl)
    set_bytes = 4.0 * wl["n"] * (2 * sh0["l"] * wl["c"] + 2 * wl["cf"] * sh0["hf"] * sh0["wf"])
    npairs = max(1, min(a.pairs, max(a.streams if wl["n"] <= 4 else 2, int(math.ceil(768e6 / set_bytes)))))
    nstreams = max(1, min(a.streams, npairs))
    pairs = []
    for p in range(npairs):      # pair p runs on stream p % nstreams and shares that stream's window buffers
        pairs.append(Pair(wl, 1000 * (rank + 1) + 17 * p, a.window, dev, a.dist,
                          share=pairs[p % nstreams] if p >= nstreams else None, layout=a.layout, fine_path=a.fine_path))
        pairs[-1].fuse_maps = pairs[-1].fuse_maps and not a.no_fuse_maps
        pairs[-1].alone = a.alone

    # Steps are independent pairs: consecutive steps go round-robin to `--streams` HIP streams so that
    # the (mostly latency-bound, small-grid) kernels of different pairs overlap on the chip.  Every input
    # set has its own buffers and its own captured graph; the timed region still covers K complete steps.
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    if a.cu_mask:
        streams = masked_streams(dev, nstreams, a.cu_mask)
    graphs = []
    for i, p in enumerate(pairs):
        with torch.cuda.stream(streams[i % nstreams]):
            p.step()
    torch.cuda.synchronize()
    for p in pairs:
        p.stages = a.stages
    if not a.no_graph:
        for i, p in enumerate(pairs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[i % nstreams]):
                p.step()
            graphs.append(g)

    def run(i):
        with torch.cuda.stream(streams[(i % npairs) % nstreams]):
            if graphs:
                graphs[i % npairs].replay()
            else:
                pairs[i % npairs].step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        run(i)
    barrier()
    # A short --steps alone is a sample of a millisecond: the timed K-step region (barrier + synchronise on both
    # sides, as the contract says) is repeated R times and the MEDIAN repetition is reported, with the spread.
    # R from a probe of the step time: >= 1.2 s of GPU work in total (long enough for a once-per-second utilisation
    # sampler outside this process to see the GPU busy), at least 3, at most 2000 repetitions.
    reps = a.reps
    if reps <= 0:
        tp = time.perf_counter()
        nprobe = max(8, min(a.steps, 64))
        for i in range(nprobe):
            run(i)
        torch.cuda.synchronize()
        est = (time.perf_counter() - tp) / nprobe
        reps = int(max(3, min(2000, math.ceil(1.2 / max(a.steps * est, 1e-6)))))
        reps = agree_max(reps, world, coll_dev)      # every rank must run the same number of repetitions
    t_region0 = time.perf_counter()
    dts, enq = timed_region(run, a.steps, reps, torch.cuda.synchronize, world, coll_dev)
    t_region = time.perf_counter() - t_region0
    dt = float(np.median(dts))
    t_enq = float(np.median(enq))

    gather_ms = gathered = None
    with torch.cuda.stream(streams[0]):
        # sanity of what was timed: the last step of every input set produced matches, no device error
        ms = []
        for p in pairs:
            ms.append(p.last[0].read_count())
        assert min(ms) > 0, "a timed step produced no matches"

        if grouped:
            # cfg#4's exchange: the match lists of the last step of EVERY resident input set, packed as 24-byte
            # records with global pair ids, gathered on every rank (dist.gather_match_lists; RCCL all-gather of
            # counts + padded records)
            recs = []
            for p, m in zip(pairs, ms):
                buf, k0, k1 = p.last
                recs.append(fdist.pack_records(buf.b_ids[:m], k0[:m, :2], k1[:m, :2], buf.mconf[:m], pair_offset=pair_lo))
            rec = torch.cat(recs).to(coll_dev)
            rec = rec[torch.argsort(fdist.unpack_records(rec)[0], stable=True)]      # pair order inside the rank
            gather_ms, gathered = timed_gather(rec, world, torch.cuda.synchronize, coll_dev)

        tk = ver = None
        if rank == 0:
            pairs[0].stages = "all"
            pairs[0].step()
            torch.cuda.synchronize()
            ver = verify(pairs[0])
            tk = time_kernels(pairs[0])

    rank_devices = None
    if grouped:
        # every rank's device identity, gathered on rank 0: a SCALE record then shows that N ranks ran on N devices
        props = torch.cuda.get_device_properties(dev)
        ident = f"rank {rank}: cuda:{dev.index} {props.name} uuid={getattr(props, 'uuid', 'n/a')} pci={getattr(props, 'pci_bus_id', 'n/a')}"
        gathered_ids = [None] * world
        dist.all_gather_object(gathered_ids, ident)
        rank_devices = gathered_ids
    if rank != 0:
        if grouped:
            dist.destroy_process_group()
        return

    pairs_per_step = wl["n"]
    value = world * a.steps * pairs_per_step / dt
    flops = 2.0 * wl["n"] * pairs[0].l * pairs[0].l * wl["c"]          # one GEMM per pair
    # the whole correlation: max pass + the sum kernels (+ the dense kernel's float16 planes when that path runs);
    # k_prep_split (the quantisation the product needs) is priced beside it
    t_corr = tk["max"] + tk["sparse"] + tk["planes"] + tk["dense"]
    ach = flops / (t_corr * 1e-3) / 1e12
    ach_max = flops / (tk["max"] * 1e-3) / 1e12
    m_avg = float(np.mean(ms)) / wl["n"]
    ww, cf = a.window * a.window, wl["cf"]
    crop_bytes = 2.0 * m_avg * wl["n"] * ww * cf * 4 * 2      # both images: read + write
    fine_bytes = 2.0 * m_avg * wl["n"] * ww * cf * 4 + 2.0 * m_avg * wl["n"] * 12
    traffic, traffic_src = committed_traffic(a.workload)
    copy_gbs = copy_rate(crop_bytes, dev)
    maps_path = pairs[0].fine_path == "maps"
    # (the common path's five: prep, max pass, thresholds, screening, assignment)
    launches = 5 + (2 if pairs[0].dense and not pairs[0].flat else 0) + \
        ((2 if a.layout == "nchw" and not pairs[0].fuse_maps else 1) if maps_path else (3 if a.layout == "nhwc" else 2))
    map_bytes = 2.0 * wl["n"] * cf * 4 * sh0["hf"] * sh0["wf"]           # both fine maps
    out = {
        "metric": ("image-pairs/sec at 640x480 (coarse corr + dual-softmax mutual-NN + fine window refinement)"
                   if a.workload == "cfg2" else f"image-pairs/sec ({a.workload})")
                  + ("" if a.stages == "all" else f" [DIAGNOSTIC: {a.stages} stage only]"),
        "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4),
        "ms_per_step_spread": {"repetitions": reps, "min": round(min(dts) / a.steps * 1e3, 4),
                               "max": round(max(dts) / a.steps * 1e3, 4), "timed_regions_total_s": round(t_region, 3),
                               "note": "the K-step region (barrier + synchronise on both sides) repeated; value and "
                                       "ms_per_step are the median repetition"},
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i8 screening + f32 exact",
        "dtype_note": "the one dense product runs on v_mfma_i32_32x32x32_i8 (screening); results are float32 (the reference's arithmetic type): int8 MFMA screening with a rigorous error "
                      "margin decides which entries matter, every entry that does gets an exact float32 dot product; the "
                      "dense fallback for flat similarity (FM_MODE_DENSE) and the context layers use hi/lo-split float16 "
                      "products, 22 significant bits",
        "data": "synthetic",
        "config": {"workload": f"{wl['label']}, {a.window}x{a.window} fine window, '{a.dist}' descriptors",
                   "pairs_per_step_per_gpu": pairs_per_step, "launch": ("eager" if a.no_graph else "hipGraph replay") + (" [DIAGNOSTIC: FM_MODE_ALONE on every step]" if a.alone else ""),
                   "concurrent_streams": nstreams, "matches_per_pair": round(m_avg, 1),
                   "launches_per_step": launches, "pair_block": [pair_lo, pair_hi],
                   "host_enqueue_ms_per_step": round(1e3 * t_enq / a.steps, 4)},
        "verified": (ver["ok"] if ver else None), "verification": ver,
        # The coarse correlation = every launch that works on the L x S product: the int8 max pass (the one dense
        # sweep) and the sparse sum kernel (re-executes the live units, exact float32 dots for the significant entries)
        # (+ float16 planes and the dense sum kernel under FM_MODE_DENSE).  Priced against the int8 MFMA peak, the
        # matrix-core type the dominant kernel runs on.
        "roofline": {"bound": "mfma",
                     "kernel": "coarse correlation: k_max_i8<256> + k_thresh + k_screen_rows<256>"
                               + (" + k_prep_f16 + k_dense<256>" if pairs[0].dense else "")
                               + " (every launch on the L x S product; v_mfma_i32_32x32x32_i8)",
                     "achieved": round(ach, 2), "peak": PEAK_I8_DENSE_TOPS, "unit": "TFLOP/s",
                     "frac": round(ach / PEAK_I8_DENSE_TOPS, 4),
                     "frac_of_f16_peak": round(ach / PEAK_F16_DENSE_TFLOPS, 4),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "avg_ms": round(t_corr, 5), "algorithmic_flop": flops,
                     "note": "algorithmic 2*L*S*C flop of the ONE product per pair over the summed event-timed launch "
                             "durations of the correlation kernels; `peak` is the dense int8 MFMA peak (2x the f16/bf16 "
                             "figure the north star names: frac_of_f16_peak)",
                     "max_pass": {"kernel": "k_max_i8<256>", "avg_ms": round(tk["max"], 5), "achieved": round(ach_max, 2),
                                  "frac": round(ach_max / PEAK_I8_DENSE_TOPS, 4),
                                  "frac_of_f16_peak": round(ach_max / PEAK_F16_DENSE_TFLOPS, 4)},
                     "sparse_sum_avg_ms": round(tk["sparse"], 5), "dense_sum_avg_ms": round(tk["dense"], 5),
                     "f16_planes_avg_ms": round(tk["planes"], 5),
                     "with_quantisation": {"k_prep_split_avg_ms": round(tk["prep"], 5),
                                           "frac": round(flops / ((t_corr + tk["prep"]) * 1e-3) / 1e12 / PEAK_I8_DENSE_TOPS, 4)},
                     "coarse_stage": {"avg_ms": round(tk["coarse"], 5),
                                      "what": "one fm_coarse_match call (all launches with their in-stream gaps, one stream)"
                                              + ("; the assignment launch also carries the channels-last copy of image 1's "
                                                 "fine map (fm_coarse_match_maps)" if pairs[0].fuse_maps else ""),
                                      "assignment_and_gaps_ms": round(tk["coarse"] - t_corr - tk["prep"], 5)}},
        "roofline_aux": ({
            "fine_from_maps": {"bound": "hbm",
                               "kernel": ("k_nchw_to_nhwc64 (image 1) + " if a.layout == "nchw" else "") + f"k_fine_maps<{a.window}> (window crop + fine stage, no window tensors)",
                               # SURVEY 8(d) fine-kernel bytes ONLY: the window bytes of both images read once + 12 bytes
                               # written per keypoint.  What the NCHW route moves beyond that (the channels-last copy of
                               # image 1: read + write of one map) is overhead, reported beside it, not work.
                               "algorithmic_bytes": fine_bytes,
                               "achieved": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "bytes_moved_by_design": fine_bytes + (map_bytes if a.layout == "nchw" else 0.0),
                               "frac_of_bytes_moved": round((fine_bytes + (map_bytes if a.layout == "nchw" else 0.0)) / (tk["fine"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "avg_ms": round(tk["fine"], 5),
                               "note": ("timed as two launches of its own (transpose + fused kernel); in the step the transpose "
                                        "rides in the coarse call's assignment launch and what follows the coarse stage is the "
                                        "fused kernel alone: after_coarse_avg_ms") if pairs[0].fuse_maps else None,
                               "after_coarse_avg_ms": round(tk.get("fine_prepared", tk["fine"]), 5)}} if maps_path else {
            "window_crop": {"bound": "hbm", "kernel": "k_gather_cellorder64 (both images, one launch)" if a.layout == "nchw"
                                                      else f"2 x k_gather_nhwc64<{a.window}> (channels-last maps: 16-byte-chunk copy)",
                            "achieved": round(crop_bytes / (tk["crop"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(crop_bytes / (tk["crop"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "avg_ms": round(tk["crop"], 5), "algorithmic_bytes": crop_bytes},
            "fine_match": {"bound": "hbm", "kernel": f"k_fine<{a.window}>",
                           "achieved": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "avg_ms": round(tk["fine"], 5), "algorithmic_bytes": fine_bytes}}) | {
            "plain_copy": {"what": "torch device-to-device copy moving as many bytes as the window crop",
                           "achieved": round(copy_gbs, 1), "unit": "GB/s", "frac": round(copy_gbs / HBM_PEAK_GBS, 4)}},
    }
    out["config"]["fine_maps_layout"] = a.layout
    out["config"]["fine_path"] = ("fm_fine_match_maps (crop + fine from the maps)" if maps_path
                                  else "window crop -> fm_fine_match")
    if grouped:
        out["gather_ms"] = round(gather_ms, 4)
        out["gathered_records"] = gathered
        # what the collective backend saw: the world size torch.distributed reports and one device identity per rank
        out["distributed"] = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""),
                              "world_size": dist.get_world_size(), "device_uuids": rank_devices}
    if world == 1 and not a.quick and a.stages == "all":
        out["extra"] = extras(a, wl, dev, streams, flops)
        # the headline is measured on 'peaky' descriptors (every conf is 1.0, one significant entry per row);
        # the same step on data with textureless cells and missing partners stands next to it
        for key, name in (("mixed_data", "value_on_mixed"), ("borderline_data", "value_on_borderline")):
            v = out["extra"].get(key, {})
            out["config"][name] = v.get("value") if isinstance(v, dict) else None
    if not a.skip_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, a.window, 1)
    print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()


def extras(a, wl, dev, streams, flops):
    """Secondary lines of the default run (each self-verified against the oracle; none can take the headline down):
    other data distributions at the metric's size, the other BASELINE configurations, the reference-shaped module
    call, the layers either side of the path."""
    extra = {}

    def guarded(name, fn):
        try:
            extra[name] = fn()
        except Exception as e:       # a secondary line must not take the headline down
            extra[name] = {"error": repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def dist_line(dist):
        """kernel times and throughput of the metric's step on another data distribution (FM_MODE_DENSE on)"""
        with torch.cuda.stream(streams[0]):
            p = Pair(wl, 7777, a.window, dev, dist)
            p.step()
            torch.cuda.synchronize()
            p.last[0].read_count()
            tb = time_kernels(p)
        tcb = tb["max"] + tb["sparse"] + tb["planes"] + tb["dense"]
        del p
        # with and without the hint, alternating, the better of two runs each: the first line after a CPU-side pause reads
        # low whatever it measures (same-process A/B: tools/time_flat.py)
        rate = rate_nohint = 0.0
        for _ in range(2):
            r1, ver, m_pp = stream_rate(wl, a.window, dev, dist, 1, 4, steps=400, nsets=8)
            r2, ver2, _ = stream_rate(wl, a.window, dev, dist, 1, 4, steps=400, nsets=8, flat_hint=False)
            rate, rate_nohint = max(rate, r1), max(rate_nohint, r2)
        return {"value": round(rate, 2), "unit": "image-pairs/s", "verified": (ver["ok"] and ver2["ok"]) if ver and ver2 else None,
                "verification": ver, "matches_per_pair": round(m_pp, 1),
                "mode": "what fm_coarse_match_auto's hint word names for this data after two calls on it (bench.learnt_hint: no "
                        "mode is hand-passed): " + str(ops.HintMemory.decode(_HINTS.get((1, synth.config_shapes(wl)["l"], wl["c"], dist), 0))),
                "value_without_flat_hint": round(rate_nohint, 2),
                "corr_avg_ms": round(tcb, 5), "frac": round(flops / (tcb * 1e-3) / 1e12 / PEAK_I8_DENSE_TOPS, 4),
                "frac_of_f16_peak": round(flops / (tcb * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4),
                "max_pass_avg_ms": round(tb["max"], 5), "sparse_sum_avg_ms": round(tb["sparse"], 5),
                "dense_sum_avg_ms": round(tb["dense"], 5), "f16_planes_avg_ms": round(tb["planes"], 5)}

    if a.workload == "cfg2" and a.batch == 0:
        guarded("borderline_data", lambda: dict(dist_line("borderline"), note=(
            "'borderline' descriptors (flat similarity): no unit is negligible, the f32-equivalent hi/lo product runs "
            "on all of them (3 f16 MFMA per k-step: ceiling 1/3 of the f16 peak); 4 streams, inputs generated on the device")))
        guarded("mixed_data", lambda: dict(dist_line("mixed"), note=(
            "'peaky' descriptors with 20 % near-zero cells in both images (textureless regions, missing partners): the "
            "near-zero rows / columns are certified dead by the sparse sum kernel (||a||_1 max|b| / (C T) bounds their "
            "softmax terms below thr) and cost nothing; the cells whose partner is missing are rows without a peak and "
            "send the sample through the dense sum kernel (FM_MODE_FLAT, 16 candidate slots, FM_MODE_EXACT_STEP instead of "
            "the exact screening pass)")))

        def cfg3_line():
            w3 = dict(WORKLOADS["cfg3"])
            rate, ver, m_pp = stream_rate(w3, a.window, dev, "peaky", w3["n"], 2, steps=6, nsets=2)
            res = {"value": round(rate, 2), "unit": "image-pairs/s", "pairs_per_step": w3["n"],
                   "verified": ver["ok"] if ver else None, "verification": ver, "matches_per_pair": round(m_pp, 1)}
            # the same batch with channels-last fine maps (no copy of image 1, no in-place NCHW window loads): what the
            # reference's NCHW hand-over costs at the batch
            rate_cl, ver_cl, _ = stream_rate(w3, a.window, dev, "peaky", w3["n"], 2, steps=6, nsets=2, layout="nhwc", fine_path="maps")
            res["channels_last_maps"] = {"value": round(rate_cl, 2), "unit": "image-pairs/s", "verified": ver_cl["ok"] if ver_cl else None}
            # the coarse correlation's roofline at the batch - the regime in which the matrix cores decide: event-timed
            # launches of the max pass and the screening kernels on a filled workspace, algorithmic 2 N L S C flop
            with torch.cuda.stream(streams[0]):
                p3 = Pair(w3, 9000, a.window, dev, "peaky")
                p3.step()
                torch.cuda.synchronize()
                p3.last[0].read_count()
                t3 = time_kernels(p3)
                del p3
            f3 = 2.0 * w3["n"] * (w3["h"] // 8 * (w3["w"] // 8)) ** 2 * w3["c"]
            tc3 = t3["max"] + t3["sparse"]
            tr3, tr3_src = committed_traffic("cfg3")
            res["roofline"] = {"bound": "mfma", "kernel": "coarse correlation: k_max_i8<256> + k_thresh + k_screen_rows<256>",
                               "achieved": round(f3 / (tc3 * 1e-3) / 1e12, 2), "peak": PEAK_I8_DENSE_TOPS, "unit": "TFLOP/s",
                               "frac": round(f3 / (tc3 * 1e-3) / 1e12 / PEAK_I8_DENSE_TOPS, 4),
                               "frac_of_f16_peak": round(f3 / (tc3 * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4),
                               "traffic": tr3, "traffic_source": tr3_src, "avg_ms": round(tc3, 5), "algorithmic_flop": f3,
                               "max_pass": {"avg_ms": round(t3["max"], 5),
                                            "frac": round(f3 / (t3["max"] * 1e-3) / 1e12 / PEAK_I8_DENSE_TOPS, 4)},
                               "screening_avg_ms": round(t3["sparse"], 5), "k_prep_split_avg_ms": round(t3["prep"], 5),
                               "coarse_stage_avg_ms": round(t3["coarse"], 5), "fine_avg_ms": round(t3["fine"], 5)}
            # materialise-conf mode (coarse_matching_new.py:70; BASELINE config 3 "HBM-bound stress"): the dense
            # [N,L,S] float32 conf_matrix is written by one more sweep
            p = Pair(w3, 9100, a.window, dev, "peaky")
            p.step()
            torch.cuda.synchronize()
            p.conf_matrix, p.dense, p.stages = True, True, "coarse"
            p.fuse_maps = False          # (the coarse stage alone: no fine-map copy on board)
            p.step()
            torch.cuda.synchronize()
            # (median of five single steps: every step allocates its 5.9 GB conf_matrix, and one step that has to go to the
            # driver for it - 40 ms, seen once - must not be the line)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                p.step()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t_ms = float(np.median(ts))
            buf = p.last[0]
            buf.read_count()
            from oracle import matcher_ref as orc     # checker only
            refc = orc.conf_matrix(p.f0[:1].cpu(), p.f1[:1].cpu(), 0.1)[0]
            got = buf.conf_matrix[0].cpu()
            cerr = float((got - refc).abs().max())
            # the sweep's hi/lo-split products carry 22 bits (2e-4 in a conf near 1 at |sim| ~ 160); every entry that
            # matters is rewritten from its exact float32 dot product (k_conf_patch): the bar is BASELINE.md's 1e-5
            smax = float((p.f0[0] @ p.f1[0].T).abs().max()) / (p.c * 0.1)
            cbar = 1e-5
            nbytes = 4.0 * w3["n"] * p.l * p.l
            res["materialise_conf_matrix"] = {
                "coarse_stage_ms": round(t_ms, 3), "conf_matrix_bytes": nbytes,
                "achieved_GBs_whole_stage": round(nbytes / (t_ms * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                "image_pairs_per_s_coarse_only": round(w3["n"] / (t_ms * 1e-3), 1),
                "conf_matrix_max_abs_err_sample0": cerr, "error_bar": cbar, "largest_abs_similarity": round(smax, 1),
                "verified": bool(cerr <= cbar),
                "note": "coarse stage with data['conf_matrix'] requested (FM_MODE_DENSE | exact screening off): prep, max "
                        "pass, screening, float16 planes, denominator reduction, dense conf sweep (k_dense<256, CONF_LITE> - one float16 product for screened samples, hi/lo-split k_dense<256, CONF> for flat ones: the "
                        "5.9 GB write) + exact rewrite of the entries that matter (k_conf_patch), assignment; GB/s = conf_matrix bytes over the WHOLE stage's time"}
            return res
        guarded("cfg3", cfg3_line)

        def cfg5_line():
            w5 = dict(WORKLOADS["cfg5"])
            rate, ver, m_pp = stream_rate(w5, a.window, dev, "peaky", 1, 4, steps=80, nsets=4)
            return {"value": round(rate, 2), "unit": "image-pairs/s", "verified": ver["ok"] if ver else None,
                    "verification": ver, "matches_per_pair": round(m_pp, 1), "workload": w5["label"]}
        guarded("cfg5", cfg5_line)

        def l9600_line():
            w9 = dict(WORKLOADS["l9600"])
            rate, ver, m_pp = stream_rate(w9, a.window, dev, "peaky", 1, 4, steps=160, nsets=6)
            return {"value": round(rate, 2), "unit": "image-pairs/s", "verified": ver["ok"] if ver else None,
                    "verification": ver, "matches_per_pair": round(m_pp, 1), "workload": w9["label"],
                    "algorithmic_flop_per_pair": 2.0 * 9600 * 9600 * 256}
        guarded("l9600", l9600_line)

        def cl_line():
            """channels-last fine maps: the crop as a 16-byte-chunk copy, and crop + fine from the maps in one kernel"""
            res = {}
            with torch.cuda.stream(streams[0]):
                p = Pair(wl, 8888, a.window, dev, "peaky", device_data=True, layout="nhwc", fine_path="windows")
                p.step()
                torch.cuda.synchronize()
                m = p.last[0].read_count()
                tw = time_kernels(p)
                wb = 2.0 * m * a.window * a.window * wl["cf"] * 4
                res["window_crop"] = {"kernel": f"2 x k_gather_nhwc64<{a.window}>", "avg_ms": round(tw["crop"], 5),
                                      "algorithmic_bytes": 2 * wb, "achieved_GBs": round(2 * wb / (tw["crop"] * 1e-3) / 1e9, 1),
                                      "frac_of_8TBs": round(2 * wb / (tw["crop"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                p.fine_path = "maps"
                p.step()
                torch.cuda.synchronize()
                tm = time_kernels(p)
                res["fine_from_maps"] = {"kernel": f"k_fine_maps<{a.window}>", "avg_ms": round(tm["fine"], 5),
                                         "algorithmic_bytes": wb + m * 24.0,
                                         "achieved_GBs": round((wb + m * 24.0) / (tm["fine"] * 1e-3) / 1e9, 1),
                                         "frac_of_8TBs": round((wb + m * 24.0) / (tm["fine"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "window bytes read once (from the maps), 12 bytes written per keypoint; "
                                                 "crop + fine_match on the same maps move 3x the window bytes"}
                del p
            rate, ver, m_pp = stream_rate(wl, a.window, dev, "peaky", 1, 4, steps=600, nsets=8, layout="nhwc", fine_path="maps")
            res.update({"value": round(rate, 2), "unit": "image-pairs/s", "verified": ver["ok"] if ver else None,
                        "verification": ver, "launches_per_step": 6,
                        "note": "the metric's step with channels-last fine maps (what a backbone that keeps its [B,H,W,C] "
                                "activations hands over): coarse stage (5 launches) + k_fine_maps; 4 streams"})
            return res
        guarded("channels_last_maps", cl_line)

        guarded("batched_launches", lambda: {
            "value": round(stream_rate(wl, a.window, dev, "peaky", 4, 4, steps=240, nsets=6, check=False)[0], 2),
            "unit": "image-pairs/s", "pairs_per_launch": 4, "concurrent_streams": 4,
            "note": "the same step with 4 pairs per launch: not the metric's configuration (one pair per step), "
                    "reported for servers that group requests"})
        guarded("one_stream", lambda: {
            "value": round(stream_rate(wl, a.window, dev, "peaky", 1, 1, steps=300, nsets=6, check=False, alone=True)[0], 2),
            "value_without_alone_hint": round(stream_rate(wl, a.window, dev, "peaky", 1, 1, steps=300, nsets=6, check=False)[0], 2),
            "unit": "image-pairs/s",
            "note": "the metric's step on ONE stream (no overlap between pairs) - the reference's only shipped caller is one "
                    "pair per call (demo/demo.py:95-116) - with FM_MODE_ALONE: the caller's hint that it has the GPU to itself "
                    "(grids sized for the kernel alone); value_without_alone_hint = the four-stream configuration's grids"})
    guarded("module_api", lambda: {"value": round(module_api_rate(wl, a.window, dev), 2), "unit": "image-pairs/s",
                                   "note": "modules.CoarseMatching -> window crop -> modules.FineMatching, one pair at a "
                                           "time, with the host sync on the match count and per-call allocations"})
    if a.workload == "cfg2":
        guarded("context_layers", lambda: context_layer_times(wl, dev))
    # correct-but-slow routes taken anywhere in this process (0 = none): the context layers' float32 torch layers behind
    # k_fine_tf's range report, the torch formula of the conf_matrix backward
    cl = extra.get("context_layers", {})
    extra["slow_paths"] = dict(ops.SLOW_PATHS, fine_tf_range_fallbacks=(cl.get("fine", {}) or {}).get("range_fallbacks")
                               if isinstance(cl, dict) else None)
    return extra


if __name__ == "__main__":
    main()
