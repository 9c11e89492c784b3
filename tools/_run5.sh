set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/pytest_r3f.log 2>&1; tail -15 gpurun_out/pytest_r3f.log
timeout -k 10 100 python tools/time_fine_tf.py > gpurun_out/time_fine_tf.log 2>&1; tail -4 gpurun_out/time_fine_tf.log
