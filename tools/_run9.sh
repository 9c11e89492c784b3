set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "maps or gather or full_path" > gpurun_out/pytest_r3j.log 2>&1 || { tail -30 gpurun_out/pytest_r3j.log; exit 1; }
tail -2 gpurun_out/pytest_r3j.log
timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 > gpurun_out/bench_r3j.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_r3j.json'));print('value',d['value'],'verified',d['verified'],'aux',{k:(v['avg_ms'],v.get('frac')) for k,v in d['roofline_aux'].items() if 'avg_ms' in v})"
