#!/usr/bin/env python3
"""Reduces gpurun_out/prof (written by tools/collect_profiles.sh on the GPU box) to the small files
committed under profiles/:

    python tools/summarize_profiles.py r01

  profiles/<round>_bench_cfg2_default_kernel_stats.csv   rocprofv3 --stats of the default bench command
  profiles/<round>_bench_cfg2_serial_kernel_stats.csv    same step, one stream, eager (no overlap)
  profiles/<round>_bench_cfg3_kernel_stats.csv, ..._cfg5_...
  profiles/<round>_bench_cfg2_borderline_serial_kernel_stats.csv, ..._mixed_...   flat-similarity data (dense path)
  profiles/<round>_conf_matrix_cfg3_kernel_stats.csv     tools/time_conf_matrix.py: the coarse stage writing data['conf_matrix'] for 64 pairs
  profiles/<round>_pmc_fetch_write_cfg2.json, ..._cfg3.json   FETCH_SIZE / WRITE_SIZE per launch and kernel (KiB)
  profiles/<round>_pmc_sq_cfg2.csv, ..._cfg3.csv         matrix-core busy / wait fractions per kernel
  profiles/<round>_forward_features_kernel_stats.csv     rocprofv3 --stats of tools/time_matcher.py (net.forward tail)
  profiles/<round>_pmc_sq_forward_features.csv           the same counters for the context-layer kernels
  profiles/<round>_bench_lines.json                      the JSON lines the profiled commands printed
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    """newest match: gpurun merges results into gpurun_out/ without removing those of earlier calls"""
    f = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void fm::", "").replace("fm::", "").split("(")[0]


def stats(tag, out):
    f = one(f"{tag}/**/*kernel_stats.csv")
    if not f:
        return
    with open(f) as fh, open(os.path.join(DST, out), "w", newline="") as oh:
        w = csv.writer(oh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for r in csv.DictReader(fh):
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                        f"{float(r['MaxNs']) / 1e3:.2f}", r["Percentage"]])
    print("wrote", out)


def counters(tag):
    f = one(f"{tag}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if f:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    stats("default", f"{rnd}_bench_cfg2_default_kernel_stats.csv")
    stats("serial", f"{rnd}_bench_cfg2_serial_kernel_stats.csv")
    stats("cfg3", f"{rnd}_bench_cfg3_kernel_stats.csv")
    stats("cfg3serial", f"{rnd}_bench_cfg3_serial_kernel_stats.csv")
    stats("cfg5", f"{rnd}_bench_cfg5_kernel_stats.csv")
    stats("borderline", f"{rnd}_bench_cfg2_borderline_serial_kernel_stats.csv")
    stats("mixed", f"{rnd}_bench_cfg2_mixed_serial_kernel_stats.csv")
    stats("conf", f"{rnd}_conf_matrix_cfg3_kernel_stats.csv")

    sha_file = os.path.join(SRC, "kernel_src_sha16.txt")       # written on the GPU box by collect_profiles.sh
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
    for tags, wl in ((("fetch", "write"), "cfg2"), (("fetch3", "write3"), "cfg3")):
        fw = {}
        for tag in tags:
            for k, cs in counters(tag).items():
                for c, v in cs.items():
                    fw.setdefault(k, {})[c] = sum(v) / len(v)
                    fw[k]["launches_" + c] = len(v)
        if fw:
            with open(os.path.join(DST, f"{rnd}_pmc_fetch_write_{wl}.json"), "w") as oh:
                json.dump({"_note": "average per launch, KiB as rocprofv3 reports them; FETCH_SIZE must be doubled "
                                    "for wide reads on gfx950 (MI355X_MICROARCH.md, HBM section); kernel_src_sha16 = "
                                    "bench.kernel_source_sha() of the sources the counters were collected with",
                           "kernel_src_sha16": sha, "kernels": fw}, oh, indent=1)
            print("wrote pmc_fetch_write", wl)

    stats("ctx", f"{rnd}_forward_features_kernel_stats.csv")
    for tag, out in (("sq", f"{rnd}_pmc_sq_cfg2.csv"), ("sq3", f"{rnd}_pmc_sq_cfg3.csv"), ("sqflat", f"{rnd}_pmc_sq_cfg2_borderline.csv"),
                     ("ctxsq", f"{rnd}_pmc_sq_forward_features.csv")):
        sq = counters(tag)
        if not sq:
            continue
        with open(os.path.join(DST, out), "w", newline="") as oh:
            w = csv.writer(oh)
            w.writerow(["kernel", "launches", "mfma_busy_frac_of_simd_cycles", "wait_any", "wait_inst_any",
                        "active_inst_any", "wait_inst_lds", "gui_active_cycles_per_launch"])
            for k, cs in sq.items():
                if "SQ_WAVE_CYCLES" not in cs:
                    continue
                n = len(cs["SQ_WAVE_CYCLES"])
                mean = {c: sum(v) / len(v) for c, v in cs.items()}
                wc = mean["SQ_WAVE_CYCLES"] or 1.0
                gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
                # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles are summed over the 1024 SIMDs
                cyc = gui / 8.0
                mfma = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc if cyc else 0.0
                w.writerow([k, n, f"{mfma:.4f}", f"{mean.get('SQ_WAIT_ANY', 0) / wc:.4f}",
                            f"{mean.get('SQ_WAIT_INST_ANY', 0) / wc:.4f}", f"{mean.get('SQ_ACTIVE_INST_ANY', 0) / wc:.4f}",
                            f"{mean.get('SQ_WAIT_INST_LDS', 0) / wc:.4f}", f"{cyc:.0f}"])
        print("wrote", out)

    lines = {}
    for tag in ("default", "serial", "fetch", "write", "sq", "cfg3", "cfg5"):
        p = os.path.join(SRC, f"{tag}.json")
        if os.path.exists(p):
            txt = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if txt:
                lines[tag] = json.loads(txt[-1])
    with open(os.path.join(DST, f"{rnd}_bench_lines.json"), "w") as oh:
        json.dump(lines, oh, indent=1)
    print("wrote bench_lines")


if __name__ == "__main__":
    main()
