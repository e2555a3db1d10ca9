set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_training.py -m gpu -x -q 2>&1 | tail -25
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/s6_train_tb.log 2>&1; tail -3 gpurun_out/s6_train_tb.log | cut -c1-1500
