set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r3c.log 2>&1; tail -3 gpurun_out/pytest_r3c.log
timeout -k 10 200 python tools/gpu_bringup.py --case small > gpurun_out/bringup_small.log 2>&1; echo "bringup small rc=$?"
timeout -k 10 200 python tools/diag_sparse.py build/variants/libfmatch_diagclock.so > gpurun_out/diag_sparse.log 2>&1; tail -12 gpurun_out/diag_sparse.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3c.json 2>gpurun_out/bench_r3c.err
python -c "
import json;d=json.load(open('gpurun_out/bench_r3c.json'));print('value',d['value'],'max_ms',d['roofline']['max_pass']['avg_ms'],'sparse',d['roofline']['sparse_sum_avg_ms'],'prep',d['roofline']['with_quantisation']['k_prep_split_avg_ms'],'coarse',d['roofline']['coarse_stage']['avg_ms'], 'verified', d['verified'])"
