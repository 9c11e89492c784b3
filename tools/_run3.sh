set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "maps or gather" > gpurun_out/pytest_r3e.log 2>&1; tail -3 gpurun_out/pytest_r3e.log
for cfg in "--layout nchw --fine-path maps" "--layout nhwc --fine-path maps"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout -k 10 300 python bench.py --quick --skip-cpu --steps 1500 $cfg > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
  python -c "
import json;d=json.load(open('gpurun_out/bench_$tag.json'));print('$tag value',d['value'],'verified',d['verified'], {k:(v['avg_ms'],v.get('frac')) for k,v in d['roofline_aux'].items() if 'avg_ms' in v})"
done
