#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence bench.py's numbers rest on.
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'
# then, back in the build container:  python tools/summarize_profiles.py rNN
# Kernel timing and PMC counters are collected in separate runs (--kernel-trace/--stats only with
# timing; --kernel-trace + --pmc only with counters), FETCH_SIZE and WRITE_SIZE in separate passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
COMMON="--steps 200 --warmup 20 --skip-cpu --quick"
python -c "import bench; print(bench.kernel_source_sha())" > $O/kernel_src_sha16.txt
# (1) the default command (hipGraph replay, 4 streams): per-kernel time under the bench's own concurrency
rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python bench.py $COMMON > $O/default.json 2> $O/default.err
echo "default done"
# (2) the same step, one stream, eager: kernel durations without overlap (these must agree with roofline.avg_ms)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python bench.py $COMMON --streams 1 --pairs 1 --no-graph > $O/serial.json 2> $O/serial.err
echo "serial done"
# (3) memory-side traffic per launch, one counter per pass
PM="--steps 20 --warmup 5 --skip-cpu --quick --streams 1 --pairs 1 --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py $PM > $O/fetch.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py $PM > $O/write.json 2> $O/write.err
echo "write done"
# (3b) the same two counters on the batch of 64 pairs (extra.cfg3.roofline.traffic)
PM3="--workload cfg3 --steps 4 --warmup 2 --skip-cpu --quick --streams 1 --pairs 1 --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch3 -- python bench.py $PM3 > $O/fetch3.json 2> $O/fetch3.err
echo "fetch3 done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write3 -- python bench.py $PM3 > $O/write3.json 2> $O/write3.err
echo "write3 done"
# (4) matrix-core utilisation of the sweeps (SQ block, one pass)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python bench.py $PM > $O/sq.json 2> $O/sq.err
echo "sq done"
# (4b) the same counters in the regime where the matrix cores decide: the batch of 64 pairs
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq3 -- python bench.py --workload cfg3 --steps 4 --warmup 2 --skip-cpu --quick --streams 1 --pairs 1 --no-graph > $O/sq3.json 2> $O/sq3.err
echo "sq3 done"
# (4c) ... and on flat data (the dense sum kernel's matrix-core utilisation)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sqflat -- python bench.py --dist borderline $PM > $O/sqflat.json 2> $O/sqflat.err
echo "sqflat done"
# (5a) the batch of 64 pairs, one stream, eager: per-kernel durations without overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3serial -- python bench.py --workload cfg3 --steps 6 --warmup 2 --skip-cpu --quick --streams 1 --pairs 1 --no-graph > $O/cfg3serial.json 2> $O/cfg3serial.err
echo "cfg3serial done"
# (5) the batch-of-64 workload (cfg3) and the 1024x1024 pair (cfg5): kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3 -- python bench.py --workload cfg3 --steps 20 --warmup 3 --skip-cpu --quick > $O/cfg3.json 2> $O/cfg3.err
echo "cfg3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -- python bench.py --workload cfg5 --steps 50 --warmup 5 --skip-cpu --quick > $O/cfg5.json 2> $O/cfg5.err
echo "cfg5 done"
# (5b) flat-similarity data (FM_MODE_DENSE | FM_MODE_FLAT): kernel time of the dense path, one stream, eager
rocprofv3 --kernel-trace --stats --output-format csv -d $O/borderline -- python bench.py --dist borderline --steps 100 --warmup 10 --skip-cpu --quick --streams 1 --pairs 1 --no-graph > $O/borderline.json 2> $O/borderline.err
echo "borderline done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mixed -- python bench.py --dist mixed --steps 100 --warmup 10 --skip-cpu --quick --streams 1 --pairs 1 --no-graph > $O/mixed.json 2> $O/mixed.err
echo "mixed done"
# (5c) BASELINE config 3's HBM-bound mode: the coarse stage with data['conf_matrix'] at 64 pairs (the 5.9 GB write)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/conf -- python tools/time_conf_matrix.py > $O/conf.log 2> $O/conf.err
echo "conf done"
# (6) the context layers either side of the path: kernel time, then matrix-core counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ctx -- python tools/time_matcher.py > $O/ctx.log 2> $O/ctx.err
echo "ctx done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/ctxsq -- python tools/time_matcher.py --hip-only > $O/ctxsq.log 2> $O/ctxsq.err
echo "ctxsq done"
# keep what is merged back small: stats + counter tables only (traces of 200 steps are large)
find $O -name '*kernel_trace.csv' -delete
find $O -name '*.db' -delete
du -sh $O
