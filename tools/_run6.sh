set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_q
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_r3g.log 2>&1; tail -6 gpurun_out/pytest_r3g.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q/cfg3 -- python bench.py --workload cfg3 --steps 12 --warmup 3 --skip-cpu --quick > gpurun_out/prof_q/cfg3.json 2> gpurun_out/prof_q/cfg3.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q/serial -- python bench.py --steps 200 --warmup 20 --skip-cpu --quick --streams 1 --pairs 1 --no-graph > gpurun_out/prof_q/serial.json 2> gpurun_out/prof_q/serial.err
find gpurun_out/prof_q -name '*kernel_trace.csv' -size +4M -delete; find gpurun_out/prof_q -name '*.db' -delete
for t in cfg3 serial; do f=$(find gpurun_out/prof_q/$t -name '*kernel_stats.csv' | head -1); echo "== $t"; head -12 $f | cut -c1-160; done
