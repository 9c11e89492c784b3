set -o pipefail
mkdir -p gpurun_out
for w in 64 128 192 256 384; do
  FM_TARGET_WGS_S=$w timeout -k 10 200 python tools/bench_variant.py build/variants/libfmatch_tune.so --quick --skip-cpu --steps 1500 --pairs 8 --fine-path maps > gpurun_out/tune_s_$w.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/tune_s_$w.json'));print('WGS_S=$w value',d['value'],'sparse',d['roofline']['sparse_sum_avg_ms'],'coarse',d['roofline']['coarse_stage']['avg_ms'])"
done
for st in 2 3 5 6 8; do
  timeout -k 10 200 python bench.py --quick --skip-cpu --steps 1500 --pairs 12 --fine-path maps --streams $st > gpurun_out/tune_st_$st.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/tune_st_$st.json'));print('streams=$st value',d['value'])"
done
