set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_q
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r3h.log 2>&1 || { tail -30 gpurun_out/pytest_r3h.log; exit 1; }
tail -3 gpurun_out/pytest_r3h.log
timeout -k 10 300 python tools/diag_conf_f64.py --workload cfg2 > gpurun_out/diag_conf_f64_cfg2.log 2>&1 || { tail -5 gpurun_out/diag_conf_f64_cfg2.log; exit 1; }
tail -4 gpurun_out/diag_conf_f64_cfg2.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3h.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3h.json'));print('value',d['value'],'coarse',d['roofline']['coarse_stage'],'prep',d['roofline']['with_quantisation'])"
