set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/s6_train_tb3.log 2>&1; tail -1 gpurun_out/s6_train_tb3.log | cut -c1-300
