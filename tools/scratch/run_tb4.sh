set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_training.py -m gpu -x -q -k "nograd or time_batched" 2>&1 | tail -25
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/s6_train_tb4.log 2>&1; tail -1 gpurun_out/s6_train_tb4.log | cut -c1-300
