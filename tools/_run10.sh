set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r3k.log 2>&1 || { tail -30 gpurun_out/pytest_r3k.log; exit 1; }
tail -2 gpurun_out/pytest_r3k.log
timeout -k 10 200 python tools/diag_max.py build/variants/libfmatch_diagclock.so > gpurun_out/diag_max.log 2>&1 || { tail -20 gpurun_out/diag_max.log; exit 1; }
tail -10 gpurun_out/diag_max.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3k.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3k.json'));print('cfg2 value',d['value'],'max',d['roofline']['max_pass'],'sparse',d['roofline']['sparse_sum_avg_ms'])"
timeout -k 10 300 python bench.py --quick --skip-cpu --workload cfg3 --steps 8 --warmup 2 --streams 1 --pairs 1 > gpurun_out/bench_r3k_cfg3.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3k_cfg3.json'));print('cfg3 1-stream value',d['value'],'max',d['roofline']['max_pass'],'sparse',d['roofline']['sparse_sum_avg_ms'],'prep',d['roofline']['with_quantisation'],'coarse',d['roofline']['coarse_stage']['avg_ms'],'aux',{k:v['avg_ms'] for k,v in d['roofline_aux'].items() if 'avg_ms' in v})"
timeout -k 10 300 python bench.py --quick --skip-cpu --workload cfg5 --steps 100 --warmup 10 > gpurun_out/bench_r3k_cfg5.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3k_cfg5.json'));print('cfg5 value',d['value'],'max',d['roofline']['max_pass'],'sparse',d['roofline']['sparse_sum_avg_ms'])"
