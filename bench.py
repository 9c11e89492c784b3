#!/usr/bin/env python3
"""Benchmark of the matching hot path (coarse correlation + dual-softmax mutual-NN assignment
-> window crop -> fine correlation + soft-argmax) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the path over one batch of synthetic feature maps that are already
resident in HBM.  Default workload = BASELINE.json configs[1]: one 640x480 pair, C=256 coarse
descriptors at 1/8 (L=S=4800), 64-d fine maps at 1/2, 5x5 fine window.  Every rank works on
its own pairs (weak scaling, no data-path collective); the timed region is bracketed by a
barrier + device synchronise and the maximum over ranks is reported.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
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

PEAK_F16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
WORKLOADS = {
    "cfg2": dict(n=1, h=480, w=640, c=256, cf=64, label="640x480 pair, C=256 @1/8 (L=S=4800), Cf=64 @1/2"),
    "cfg3": dict(n=64, h=480, w=640, c=256, cf=64, label="batch of 64 640x480 pairs, C=256 @1/8"),
    "cfg5": dict(n=1, h=1024, w=1024, c=256, cf=64, label="1024x1024 pair, C=256 @1/8 (L=S=16384)"),
    "cfg1": dict(n=1, h=128, w=128, c=64, cf=64, label="128x128 pair, C=64 @1/8 (L=S=256)"),
}


class Pair:
    """Device-resident inputs of one batch of pairs plus the launch of one step on them.  The window
    buffers belong to the stream the pair runs on (`share`: pairs of one stream run one after the other)."""

    def __init__(self, wl, seed, window, dev, dist, share=None):
        sh = synth.config_shapes(wl)
        self.n, self.l, self.c = wl["n"], sh["l"], wl["c"]
        self.hw_c, self.hw_f, self.hw_i = (sh["hc"], sh["wc"]), (sh["hf"], sh["wf"]), (wl["h"], wl["w"])
        self.window = window
        f0, f1 = synth.coarse_descriptors(seed, self.n, self.l, self.c, dist)
        self.f0, self.f1 = torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev)
        if self.n <= 4:
            ff0, ff1 = synth.fine_maps(seed, self.n, wl["cf"], sh["hf"], sh["wf"])
            self.ff0, self.ff1 = torch.as_tensor(ff0, device=dev), torch.as_tensor(ff1, device=dev)
        else:   # large batches: same statistics, generated on the device (numpy hashing of 1e9 values is slow)
            g = torch.Generator(device=dev).manual_seed(seed)
            self.ff0 = torch.randn(self.n, wl["cf"], sh["hf"], sh["wf"], device=dev, generator=g)
            self.ff1 = torch.randn(self.n, wl["cf"], sh["hf"], sh["wf"], device=dev, generator=g)
        w0, b0, w1, b1 = synth.mix_weights(seed, window * window)
        self.mix0 = torch.as_tensor(np.concatenate([w0, [b0]]).astype(np.float32), device=dev)
        self.mix1 = torch.as_tensor(np.concatenate([w1, [b1]]).astype(np.float32), device=dev)
        self.cap = self.n * self.l
        if share is None:
            self.win0 = torch.empty(self.cap, window * window, wl["cf"], device=dev)
            self.win1 = torch.empty_like(self.win0)
        else:
            self.win0, self.win1 = share.win0, share.win1
        self.last = None
        self.gather = os.environ.get("FM_GATHER", "cells")      # cells | list (see ops.gather_windows)
        self.stages = "all"      # diagnostic only (--stages): "coarse" or "fine" time a part of the step

    def step(self):
        """Enqueue the whole path; nothing synchronises the host (the match count stays on the
        device and the window/fine kernels read it there)."""
        w = self.window
        if self.stages == "fine" and self.last is not None:
            buf = self.last[0]
        else:
            buf = ops.coarse_match_async(self.f0, self.f1, self.hw_c, self.hw_c, self.hw_i[0] / self.hw_c[0],
                                         cap=self.cap)
        if self.stages == "coarse" and self.last is not None:
            self.last = (buf,) + self.last[1:]
            return self.last
        if self.gather == "cells":      # both images' crops in one launch, cell order
            ops.gather_windows_pair(self.ff0, self.ff1, buf.b_ids, buf.i_ids, buf.j_ids, w, 4, self.hw_c, self.hw_c,
                                    buf.cell_maps(), count=buf.count, out0=self.win0, out1=self.win1)
        else:                           # "list": one list-ordered launch per image
            ops.gather_windows(self.ff0, buf.b_ids, buf.i_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win0)
            ops.gather_windows(self.ff1, buf.b_ids, buf.j_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win1)
        k0, k1 = ops.fine_match(self.win0, self.win1, self.mix0, self.mix1, buf.mkpts0_c, buf.mkpts1_c,
                                self.hw_i[0] / self.hw_f[0], count=buf.count)
        self.last = (buf, k0, k1)
        return self.last


def time_corr_kernel(pair, mode, iters=30):
    """Average duration (ms) of ONE launch of a coarse kernel (mode 0 = max pass, 1 = dense sum kernel,
    "sparse" = sparse sum kernel), bracketed by events on the
    stream it is launched on (torch's current stream).  All iterations are enqueued before the host
    waits, so each bracket holds the kernel and not the idle-queue launch latency of a lone dispatch
    (that reads ~6 us longer than the kernel's duration in a rocprofv3 trace)."""
    lib = _lib.load()
    buf = pair.last[0]
    ws = buf.workspace
    ptr = C.c_void_p(ws.data_ptr() + ((-ws.data_ptr()) % 256))
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    evs = []
    for _ in range(iters + 3):
        _lib.check(lib.fm_debug_reset_counters(ptr, pair.n, pair.l, pair.l, pair.c, slots, st), "reset")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if mode == "sparse":
            _lib.check(lib.fm_debug_launch_sum_sparse(ptr, C.c_void_p(pair.f0.data_ptr()), C.c_void_p(pair.f1.data_ptr()),
                                                      pair.n, pair.l, pair.l, pair.c, slots, 0.1, 0.2, st), "sparse")
        else:
            _lib.check(lib.fm_debug_launch_corr(ptr, pair.n, pair.l, pair.l, pair.c, slots, 0.1, 0.2, mode, st), "corr")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs[3:]) / iters


def time_corr_kernel_dense(pair, iters=30):
    """The dense sum kernel on the units the sparse kernel flagged (the flags and the unit count stay in the
    workspace between launches; only the candidate counters are cleared)."""
    lib = _lib.load()
    ws = pair.last[0].workspace
    ptr = C.c_void_p(ws.data_ptr() + ((-ws.data_ptr()) % 256))
    slots = lib.fm_default_cand_slots(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    evs = []
    for _ in range(iters + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.fm_debug_launch_corr(ptr, pair.n, pair.l, pair.l, pair.c, slots, 0.1, 0.2, 1, st), "corr")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in evs[3:]) / iters


def pmc_traffic_bytes(a):
    """HBM/fabric bytes per launch of the sum pass from the committed rocprofv3 PMC passes of this
    very command (profiles/r01_pmc_fetch_write_cfg2.json: separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of `bench.py --no-graph`), with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE
    counts half of the bytes of wide reads, both counters are in KiB.  None for other workloads."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_cfg2.json")
    if a.workload != "cfg2" or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    for name, c in d.items():
        if "k_corr<256, 1>" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    return None


def cpu_baseline(wl, window, seed, budget_s=12.0):
    """The CPU oracle (a port of the reference's torch ops, pinned to the reference by the golden
    fixtures) on this host's cores, on a bounded sample of the same workload."""
    from oracle import matcher_ref as orc     # cpu_baseline leg only
    sh = synth.config_shapes(wl)
    n = min(wl["n"], 1)
    f0, f1 = synth.coarse_descriptors(seed, n, sh["l"], wl["c"], "peaky")
    ff0, ff1 = synth.fine_maps(seed, n, wl["cf"], sh["hf"], sh["wf"])
    mix = synth.mix_weights(seed, window * window)
    # torch's default (all 256 hardware threads of the host) is 4x slower than a few cores for these
    # memory-bound dense passes; 8 threads was the fastest of {8,16,32,64,128} on the MI355X host
    threads = min(8, os.cpu_count() or 8)
    torch.set_num_threads(threads)
    orc.match_features(f0, f1, ff0, ff1, (wl["h"], wl["w"]), mix, w=window)     # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < 3 or (time.perf_counter() - t0 < budget_s and done < 200):
        orc.match_features(f0, f1, ff0, ff1, (wl["h"], wl["w"]), mix, w=window)
        done += n
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "image-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} x ({wl['label']}, {window}x{window} window), oracle.match_features "
                      f"(torch-CPU ops mirroring the reference, window crop without the full unfold), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--window", type=int, default=5, choices=[5, 7])
    ap.add_argument("--dist", default="peaky", choices=["peaky", "borderline"])
    ap.add_argument("--pairs", type=int, default=12,
                    help="distinct resident input sets cycled through (12 x 59 MB of inputs: far beyond the 256 MB "
                         "Infinity Cache, so every step reads its inputs from HBM)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the independent steps are spread over")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--stages", default="all", choices=["all", "coarse", "fine"],
                    help="diagnostic: time only a part of the step (the JSON line is then not the metric)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FM_BENCH_BACKEND", "nccl")     # "gloo" to rehearse N ranks on one GPU
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    wl = WORKLOADS[a.workload]
    # enough distinct input sets to exceed the Infinity Cache several times over, not more (generating them
    # with the portable hash RNG is the slow part of the set-up)
    sh0 = synth.config_shapes(wl)
    set_bytes = 4.0 * wl["n"] * (2 * sh0["l"] * wl["c"] + 2 * wl["cf"] * sh0["hf"] * sh0["wf"])
    npairs = max(1, min(a.pairs, max(a.streams if wl["n"] <= 4 else 2, int(math.ceil(768e6 / set_bytes)))))
    nstreams = max(1, min(a.streams, npairs))
    pairs = []
    for p in range(npairs):      # pair p runs on stream p % nstreams and shares that stream's window buffers
        pairs.append(Pair(wl, 1000 * (rank + 1) + 17 * p, a.window, dev, a.dist,
                          share=pairs[p % nstreams] if p >= nstreams else None))

    # Steps are independent pairs: consecutive steps go round-robin to `--streams` HIP streams so that
    # the (mostly latency-bound, small-grid) kernels of different pairs overlap on the chip.  Every input
    # set has its own buffers and its own captured graph; the timed region still covers K complete steps.
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
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

    for i in range(a.warmup):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    with torch.cuda.stream(streams[0]):
        # sanity of what was timed: the last step of every input set produced matches, no device error
        ms = []
        for p in pairs:
            ms.append(p.last[0].read_count())
        assert min(ms) > 0, "a timed step produced no matches"

        t_a = t_b = None
        if rank == 0:
            pairs[0].stages = "all"
            pairs[0].step()
            torch.cuda.synchronize()
            t_a = time_corr_kernel(pairs[0], 0)
            t_s = time_corr_kernel(pairs[0], "sparse")
            # the dense sum kernel runs after the sparse one has flagged its units (none on 'peaky' data)
            _lib.check(_lib.load().fm_debug_reset_counters(
                C.c_void_p(pairs[0].last[0].workspace.data_ptr() + ((-pairs[0].last[0].workspace.data_ptr()) % 256)),
                pairs[0].n, pairs[0].l, pairs[0].l, pairs[0].c, _lib.load().fm_default_cand_slots(0.2),
                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "reset")
            _lib.check(_lib.load().fm_debug_launch_sum_sparse(
                C.c_void_p(pairs[0].last[0].workspace.data_ptr() + ((-pairs[0].last[0].workspace.data_ptr()) % 256)),
                C.c_void_p(pairs[0].f0.data_ptr()), C.c_void_p(pairs[0].f1.data_ptr()), pairs[0].n, pairs[0].l, pairs[0].l,
                pairs[0].c, _lib.load().fm_default_cand_slots(0.2), 0.1, 0.2,
                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sparse")
            t_b = time_corr_kernel_dense(pairs[0])

    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    pairs_per_step = wl["n"]
    value = world * a.steps * pairs_per_step / dt
    flops = 2.0 * wl["n"] * pairs[0].l * pairs[0].l * wl["c"]          # one GEMM per pair
    t_corr = t_a + t_s + t_b               # the whole correlation: max pass + sparse sum + dense sum
    ach_b = flops / (t_corr * 1e-3) / 1e12
    out = {
        "metric": ("image-pairs/sec at 640x480 (coarse corr + dual-softmax mutual-NN + fine window refinement)"
                   if a.workload == "cfg2" else f"image-pairs/sec ({a.workload})")
                  + ("" if a.stages == "all" else f" [DIAGNOSTIC: {a.stages} stage only]"),
        "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 hi+lo split operands, f32 accumulate (f32-equivalent product)",
        "data": "synthetic",
        "config": {"workload": f"{wl['label']}, {a.window}x{a.window} fine window, '{a.dist}' descriptors",
                   "pairs_per_step_per_gpu": pairs_per_step, "launch": "eager" if a.no_graph else "hipGraph replay",
                   "concurrent_streams": nstreams,
                   "matches_per_pair": round(float(np.mean(ms)) / wl["n"], 1)},
        "roofline": {"bound": "mfma",
                     "kernel": "coarse correlation = k_corr<256,0> (max pass) + k_sum_sparse<256> + k_corr<256,1> (dense sum)",
                     "achieved": round(ach_b, 2), "peak": PEAK_F16_DENSE_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach_b / PEAK_F16_DENSE_TFLOPS, 4), "traffic": None,
                     "avg_ms": round(t_corr, 5), "algorithmic_flop": flops,
                     "max_pass_avg_ms": round(t_a, 5), "sparse_sum_avg_ms": round(t_s, 5),
                     "dense_sum_avg_ms": round(t_b, 5),
                     "max_pass_frac": round(flops / (t_a * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4)},
    }
    if not a.skip_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, a.window, 1)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
