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
}


class Pair:
    """Device-resident inputs of one batch of pairs plus the launch of one step on them.  The window
    buffers belong to the stream the pair runs on (`share`: pairs of one stream run one after the other)."""

    def __init__(self, wl, seed, window, dev, dist, share=None):
        sh = synth.config_shapes(wl)
        self.seed, self.dist, self.wl = seed, dist, wl
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
        self.mix = (w0, b0, w1, b1)
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
        self.crop(buf)
        k0, k1 = self.fine(buf)
        self.last = (buf, k0, k1)
        return self.last

    def crop(self, buf):
        w = self.window
        if self.gather == "cells":      # both images' crops in one launch, cell order
            ops.gather_windows_pair(self.ff0, self.ff1, buf.b_ids, buf.i_ids, buf.j_ids, w, 4, self.hw_c, self.hw_c,
                                    buf.cell_maps(), count=buf.count, out0=self.win0, out1=self.win1)
        else:                           # "list": one list-ordered launch per image
            ops.gather_windows(self.ff0, buf.b_ids, buf.i_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win0)
            ops.gather_windows(self.ff1, buf.b_ids, buf.j_ids, w, 4, self.hw_c[1], count=buf.count, out=self.win1)

    def fine(self, buf):
        return ops.fine_match(self.win0, self.win1, self.mix0, self.mix1, buf.mkpts0_c, buf.mkpts1_c,
                              self.hw_i[0] / self.hw_f[0], count=buf.count)


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
    """Event-timed launches of the kernels of one step on a workspace the step has filled: the three coarse
    correlation kernels (max pass, sparse sum, dense sum), the window crop and the fine kernel."""
    lib = _lib.load()
    buf = pair.last[0]
    ws = buf.workspace
    ptr = C.c_void_p(ws.data_ptr() + ((-ws.data_ptr()) % 256))
    slots = lib.fm_default_cand_slots(0.2)
    f0, f1 = C.c_void_p(pair.f0.data_ptr()), C.c_void_p(pair.f1.data_ptr())
    shape = (pair.n, pair.l, pair.l, pair.c, slots)

    def st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def reset():
        _lib.check(lib.fm_debug_reset_counters(ptr, *shape, st()), "reset")

    def sparse():
        _lib.check(lib.fm_debug_launch_sum_sparse(ptr, f0, f1, *shape, 0.1, 0.2, st()), "sparse")

    t = {}
    t["max"] = _events(lambda: _lib.check(lib.fm_debug_launch_corr(ptr, *shape, 0.1, 0.2, 0, st()), "max"))
    # (4 launches per reset: a row's 8 candidate slots take the one candidate each launch adds on 'peaky' data)
    t["sparse"] = _events(sparse, before=reset, group=4)
    # the dense sum kernel redoes the samples the sparse one flagged (none on 'peaky' data: it exits at once);
    # the untimed part of every iteration clears the counters and lets the sparse kernel flag again
    def reflag():
        reset()
        sparse()
    dense = lambda: _lib.check(lib.fm_debug_launch_corr(ptr, *shape, 0.1, 0.2, 1, st()), "dense")
    # float16 planes of the flagged samples (exits at once on 'peaky' data, like the dense kernel)
    reflag()
    t["planes"] = _events(lambda: _lib.check(lib.fm_debug_launch_prep_f16(ptr, f0, f1, *shape, 0, st()), "planes"))
    if pair.dist == "borderline":      # every launch redoes the pair and adds its candidates: one launch per reflag
        t["dense"] = _events(dense, before=reflag, group=1, iters=20)
    else:                              # nothing flagged: the launches exit at once and leave nothing behind
        reflag()
        t["dense"] = _events(dense)
    # restore a consistent workspace for the crop / fine timings below
    pair.stages = "all"
    pair.step()
    torch.cuda.synchronize()
    buf = pair.last[0]
    t["crop"] = _events(lambda: pair.crop(buf))
    t["fine"] = _events(lambda: pair.fine(buf))
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


def committed_traffic(workload):
    """Fabric/HBM bytes per launch of the max pass (the roofline kernel) from the committed rocprofv3 PMC passes of this
    round (profiles/r02_pmc_fetch_write_cfg2.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of `bench.py
    --streams 1 --pairs 1 --no-graph`), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half
    of the bytes of wide reads; both counters in KiB).  A constant read from that file, not measured by this
    run - hence the file name next to it; None when the profile is missing or for another workload."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_fetch_write_cfg2.json")
    if workload != "cfg2" or not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    tot = 0.0
    for name, c in d.items():
        if "k_max_i8" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            tot += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    return (int(tot) if tot else None), "profiles/r02_pmc_fetch_write_cfg2.json"


def verify(pair):
    """Compare what the timed path produced for one input set with the CPU oracle (same tolerances as the parity
    tests): identical (b,i,j) outside the guard band |conf - thr| < 2e-5, mconf within 1e-5, fine keypoints
    within 1e-3 px."""
    from oracle import matcher_ref as orc     # checker only
    if pair.n > 2:
        return None
    f0, f1 = synth.coarse_descriptors(pair.seed, pair.n, pair.l, pair.c, pair.dist)
    sh = synth.config_shapes(pair.wl)
    ff0, ff1 = synth.fine_maps(pair.seed, pair.n, pair.wl["cf"], sh["hf"], sh["wf"])
    ref = orc.match_features(f0, f1, ff0, ff1, pair.hw_i, pair.mix, w=pair.window)
    buf, k0, k1 = pair.last
    m = buf.read_count()
    got = {k: v.cpu().numpy() for k, v in buf.sliced(m).items()}
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
        return False
    conf_err = float(np.abs(got["mconf"][gi] - rconf[ri]).max())
    fine_err = max(float(np.abs(k0.cpu().numpy()[gi] - ref["mkpts0_f"].numpy()[ri]).max()),
                   float(np.abs(k1.cpu().numpy()[gi] - ref["mkpts1_f"].numpy()[ri]).max()))
    ok = not stray and conf_err <= 1e-5 and fine_err <= 1e-3 and abs(len(gk) - len(rk)) <= 4
    return {"ok": bool(ok), "matches": m, "oracle_matches": len(rk), "mconf_err": conf_err, "fine_err_px": fine_err}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(wl, window, seed, budget_s=10.0):
    """The CPU oracle (a port of the reference's torch ops, pinned to the reference by the golden fixtures) on
    this host's cores, on a bounded sample of the same workload: one row with a single thread and one with the
    thread count that measured fastest on the MI355X host (torch's default of all hardware threads is 4x slower
    for these memory-bound dense passes; 8 was the fastest of {8,16,32,64,128})."""
    from oracle import matcher_ref as orc     # cpu_baseline leg only
    sh = synth.config_shapes(wl)
    n = min(wl["n"], 1)
    f0, f1 = synth.coarse_descriptors(seed, n, sh["l"], wl["c"], "peaky")
    ff0, ff1 = synth.fine_maps(seed, n, wl["cf"], sh["hf"], sh["wf"])
    mix = synth.mix_weights(seed, window * window)
    cores = os.cpu_count() or 8
    rows = {}
    for threads, share in ((min(8, cores), 0.7), (1, 0.3)):
        torch.set_num_threads(threads)
        orc.match_features(f0, f1, ff0, ff1, (wl["h"], wl["w"]), mix, w=window)     # warm-up
        t0 = time.perf_counter()
        done = 0
        while done < 2 or (time.perf_counter() - t0 < budget_s * share and done < 200):
            orc.match_features(f0, f1, ff0, ff1, (wl["h"], wl["w"]), mix, w=window)
            done += n
        rows[threads] = (done / (time.perf_counter() - t0), done)
    best = max(rows, key=lambda k: rows[k][0])
    return {"value": round(rows[best][0], 3), "unit": "image-pairs/s", "cores": best, "kind": "port",
            "host_cores": cores, "cpu_model": cpu_model(),
            "single_thread_value": round(rows[1][0], 3),
            "sample": f"{rows[best][1]} x ({wl['label']}, {window}x{window} window) with {best} threads and "
                      f"{rows[1][1]} x with 1 thread: oracle.match_features (torch-CPU ops mirroring the reference; "
                      f"its window crop reads only the matched cells - cheaper than the reference's full F.unfold)"}


def batched_rate(wl, window, dev, batch, nstreams, steps=240, nsets=6):
    """Pairs/s of the same step with `batch` pairs per launch (the kernels take N > 1 natively) on `nstreams` streams:
    what a server that groups requests gets - fewer, larger launches amortise the per-kernel ramp and tail."""
    wb = dict(wl, n=batch)
    pairs = []
    for p in range(nsets):
        pairs.append(Pair(wb, 5000 + 31 * p, window, dev, "peaky", share=pairs[p % nstreams] if p >= nstreams else None))
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

    for i in range(24):
        run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.cuda.stream(streams[0]):
        ms = [p.last[0].read_count() for p in pairs]
    assert min(ms) > 0
    return batch * steps / dt


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
    os.environ["FM_HIP_COARSE_TF"] = os.environ["FM_HIP_FINE_TF"] = "0"
    try:
        t_c_t, _ = timed(lambda: m.coarse(x0, x1))
        t_f_t, _ = timed(lambda: m.fine(w0, w1))
    finally:
        del os.environ["FM_HIP_COARSE_TF"], os.environ["FM_HIP_FINE_TF"]
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
    res["forward_features"] = {"ms": round(t_all, 4), "image_pairs_per_s": round(1e3 * n / t_all, 1),
                               "note": "net.forward after the backbone: coarse context layers -> coarse matching -> "
                                       "crop + context merge -> fine context layers -> fine matching, eager, one pair "
                                       "per call, host sync on the match count"}
    return res


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
    ap.add_argument("--quick", action="store_true", help="skip the secondary lines (dense data, module API)")
    ap.add_argument("--batch", type=int, default=0,
                    help="diagnostic: pairs per launch (overrides the workload's batch; the JSON line is then not the metric's config)")
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
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (diagnostic: host- or GPU-bound?)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    gather_ms = gathered = None
    with torch.cuda.stream(streams[0]):
        # sanity of what was timed: the last step of every input set produced matches, no device error
        ms = []
        for p in pairs:
            ms.append(p.last[0].read_count())
        assert min(ms) > 0, "a timed step produced no matches"

        if world > 1:
            # cfg#4's exchange: the match lists of the last step, packed as 24-byte records with global pair ids,
            # gathered on every rank (dist.gather_match_lists; RCCL all-gather of counts + padded records)
            buf, k0, k1 = pairs[0].last
            m = ms[0]
            rec = fdist.pack_records(buf.b_ids[:m], k0[:m, :2], k1[:m, :2], buf.mconf[:m], pair_offset=pair_lo)
            rec = rec.to(coll_dev)
            full = fdist.gather_match_lists(rec)      # warm-up (communicator set-up)
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            for _ in range(10):
                full = fdist.gather_match_lists(rec)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) / 10 * 1e3
            gathered = int(full.shape[0])
            ids = fdist.unpack_records(full)[0]
            assert bool((ids[1:] >= ids[:-1]).all()), "gathered records are not in pair order"

        tk = ver = None
        if rank == 0:
            pairs[0].stages = "all"
            pairs[0].step()
            torch.cuda.synchronize()
            ver = verify(pairs[0])
            tk = time_kernels(pairs[0])

    if world > 1:
        t = torch.tensor([dt, gather_ms], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gather_ms = float(t[0].item()), float(t[1].item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    pairs_per_step = wl["n"]
    value = world * a.steps * pairs_per_step / dt
    flops = 2.0 * wl["n"] * pairs[0].l * pairs[0].l * wl["c"]          # one GEMM per pair
    # the whole correlation: max pass + the two sum kernels + the dense kernel's float16 planes
    t_corr = tk["max"] + tk["sparse"] + tk["planes"] + tk["dense"]
    ach = flops / (t_corr * 1e-3) / 1e12
    m_avg = float(np.mean(ms)) / wl["n"]
    ww, cf = a.window * a.window, wl["cf"]
    crop_bytes = 2.0 * m_avg * wl["n"] * ww * cf * 4 * 2      # both images: read + write
    fine_bytes = 2.0 * m_avg * wl["n"] * ww * cf * 4 + 2.0 * m_avg * wl["n"] * 12
    traffic, traffic_src = committed_traffic(a.workload)
    copy_gbs = copy_rate(crop_bytes, dev)
    out = {
        "metric": ("image-pairs/sec at 640x480 (coarse corr + dual-softmax mutual-NN + fine window refinement)"
                   if a.workload == "cfg2" else f"image-pairs/sec ({a.workload})")
                  + ("" if a.stages == "all" else f" [DIAGNOSTIC: {a.stages} stage only]"),
        "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "dtype_note": "results are float32 (the reference's arithmetic type): int8 MFMA screening with a rigorous error "
                      "margin decides which entries matter, every entry that does gets an exact float32 product",
        "data": "synthetic",
        "config": {"workload": f"{wl['label']}, {a.window}x{a.window} fine window, '{a.dist}' descriptors",
                   "pairs_per_step_per_gpu": pairs_per_step, "launch": "eager" if a.no_graph else "hipGraph replay",
                   "concurrent_streams": nstreams, "matches_per_pair": round(m_avg, 1),
                   "launches_per_step": 9, "pair_block": [pair_lo, pair_hi],
                   "host_enqueue_ms_per_step": round(1e3 * t_enq / a.steps, 4)},
        "verified": (ver["ok"] if ver else None), "verification": ver,
        # the dominant kernel of the coarse correlation: the int8 max pass (the one dense sweep; the sum kernels
        # re-execute only the live units).  `coarse_correlation` below prices all three launches of the product.
        "roofline": {"bound": "mfma", "kernel": "k_max_i8<256> (max pass: all-pairs screening product, v_mfma_i32_32x32x32_i8)",
                     "achieved": round(flops / (tk["max"] * 1e-3) / 1e12, 2), "peak": PEAK_F16_DENSE_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flops / (tk["max"] * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4),
                     "frac_of_i8_peak": round(flops / (tk["max"] * 1e-3) / 1e12 / PEAK_I8_DENSE_TOPS, 4),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "avg_ms": round(tk["max"], 5), "algorithmic_flop": flops,
                     "note": "algorithmic 2*L*S*C flop of the ONE product per pair over this kernel's event-timed launch "
                             "duration; `peak` is the dense f16/bf16 MFMA peak the north star names, `frac_of_i8_peak` "
                             "prices the same work against the int8 MFMA peak the kernel actually runs on",
                     "coarse_correlation": {
                         "kernels": "k_max_i8 + k_sum_sparse + k_corr<256,1> (dense sum kernel; exits at once when the "
                                    "sparse one flagged nothing)",
                         "avg_ms": round(t_corr, 5), "achieved": round(ach, 2), "frac": round(ach / PEAK_F16_DENSE_TFLOPS, 4),
                         "max_pass_avg_ms": round(tk["max"], 5), "sparse_sum_avg_ms": round(tk["sparse"], 5),
                         "dense_sum_avg_ms": round(tk["dense"], 5), "f16_planes_avg_ms": round(tk["planes"], 5)}},
        "roofline_aux": {
            "window_crop": {"bound": "hbm", "kernel": "k_gather_cellorder64 (both images, one launch)",
                            "achieved": round(crop_bytes / (tk["crop"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(crop_bytes / (tk["crop"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "avg_ms": round(tk["crop"], 5), "algorithmic_bytes": crop_bytes},
            "fine_match": {"bound": "hbm", "kernel": f"k_fine<{a.window}>",
                           "achieved": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(fine_bytes / (tk["fine"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "avg_ms": round(tk["fine"], 5), "algorithmic_bytes": fine_bytes},
            "plain_copy": {"what": "torch device-to-device copy moving as many bytes as the window crop",
                           "achieved": round(copy_gbs, 1), "unit": "GB/s", "frac": round(copy_gbs / HBM_PEAK_GBS, 4)}},
    }
    if world > 1:
        out["gather_ms"] = round(gather_ms, 4)
        out["gathered_records"] = gathered
    if world == 1 and not a.quick and a.stages == "all":
        extra = {}
        try:       # dense data: every unit alive, the dense sum kernel redoes the pair
            with torch.cuda.stream(streams[0]):
                p = Pair(wl, 7777, a.window, dev, "borderline")
                p.step()
                torch.cuda.synchronize()
                p.last[0].read_count()
                tb = time_kernels(p)
            tcb = tb["max"] + tb["sparse"] + tb["planes"] + tb["dense"]
            extra["borderline_data"] = {
                "corr_avg_ms": round(tcb, 5), "frac": round(flops / (tcb * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4),
                "max_pass_avg_ms": round(tb["max"], 5), "sparse_sum_avg_ms": round(tb["sparse"], 5),
                "dense_sum_avg_ms": round(tb["dense"], 5), "f16_planes_avg_ms": round(tb["planes"], 5),
                "note": "'borderline' descriptors (flat similarity): no unit is negligible, the f32-equivalent "
                        "hi/lo product runs on all of them (3 f16 MFMA per k-step: ceiling 1/3 of the f16 peak)"}
            del p
        except Exception as e:       # a secondary line must not take the headline down
            extra["borderline_data"] = {"error": repr(e)}
        if a.workload == "cfg2" and a.batch == 0:
            try:
                extra["batched_launches"] = {
                    "value": round(batched_rate(wl, a.window, dev, 4, 4), 2), "unit": "image-pairs/s",
                    "pairs_per_launch": 4, "concurrent_streams": 4,
                    "note": "the same step with 4 pairs per launch: not the metric's configuration (one pair per step), "
                            "reported for servers that group requests"}
            except Exception as e:
                extra["batched_launches"] = {"error": repr(e)}
        try:
            extra["module_api"] = {"value": round(module_api_rate(wl, a.window, dev), 2), "unit": "image-pairs/s",
                                   "note": "modules.CoarseMatching -> window crop -> modules.FineMatching, one pair at a "
                                           "time, with the host sync on the match count and per-call allocations"}
        except Exception as e:
            extra["module_api"] = {"error": repr(e)}
        if a.workload == "cfg2":
            try:
                extra["context_layers"] = context_layer_times(wl, dev)
            except Exception as e:
                extra["context_layers"] = {"error": repr(e)}
        out["extra"] = extra
    if not a.skip_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, a.window, 1)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
