set -x
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "split or packed" 2>&1 | tail -5
python tools/microbench_chain.py 64 2>&1 | grep -E "linear|load128" > gpurun_out/s6_micro_fp32.log
TBX_SPLIT_BF16=1 python tools/microbench_chain.py 64 2>&1 | grep -E "linear|load128" > gpurun_out/s6_micro_split.log
paste gpurun_out/s6_micro_fp32.log gpurun_out/s6_micro_split.log
python bench.py --no-cpu-baseline --no-train-shape > gpurun_out/s6_bench_fp32.log 2>&1; tail -1 gpurun_out/s6_bench_fp32.log | cut -c1-600
TBX_SPLIT_BF16=1 python bench.py --no-cpu-baseline --no-train-shape > gpurun_out/s6_bench_split.log 2>&1; tail -1 gpurun_out/s6_bench_split.log | cut -c1-600
