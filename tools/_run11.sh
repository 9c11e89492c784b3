set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r3l.log 2>&1 || { tail -30 gpurun_out/pytest_r3l.log; exit 1; }
tail -2 gpurun_out/pytest_r3l.log
timeout -k 10 200 python tools/gpu_bringup.py --case small > gpurun_out/bringup_small.log 2>&1 || { tail -20 gpurun_out/bringup_small.log; exit 1; }
timeout -k 10 200 python tools/diag_sparse.py build/variants/libfmatch_diagclock.so > gpurun_out/diag_sparse.log 2>&1 || { tail -20 gpurun_out/diag_sparse.log; exit 1; }
tail -12 gpurun_out/diag_sparse.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3l.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3l.json'));print('cfg2 value',d['value'],'verified',d['verified'],'max',d['roofline']['max_pass']['avg_ms'],'sparse',d['roofline']['sparse_sum_avg_ms'],'coarse',d['roofline']['coarse_stage']['avg_ms'],'frac',d['roofline']['frac'])"
