set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r3i.log 2>&1 || { tail -30 gpurun_out/pytest_r3i.log; exit 1; }
tail -2 gpurun_out/pytest_r3i.log
timeout -k 10 200 python tools/diag_select.py build/variants/libfmatch_diagclock.so > gpurun_out/diag_select.log 2>&1 || { tail -20 gpurun_out/diag_select.log; exit 1; }
tail -20 gpurun_out/diag_select.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3i.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3i.json'));print('value',d['value'],'coarse',d['roofline']['coarse_stage'],'prep',d['roofline']['with_quantisation'])"
