#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
run() { python bench.py --no-cpu-baseline --no-wosac-shape --profile-steps 0 --new-scenes 0 --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for lm in 384 0; do
  echo "TBX_LIVE_MAX=$lm  wosac fp32:"; TBX_LIVE_MAX=$lm run --agents 128 --rollouts 32
  echo "TBX_LIVE_MAX=$lm  wosac reduced:"; TBX_LIVE_MAX=$lm run --agents 128 --rollouts 32 --kv-bf16 --attn-mfma 1
  echo "TBX_LIVE_MAX=$lm  submission fp32:"; TBX_LIVE_MAX=$lm run --agents 128 --rollouts 128
  echo "TBX_LIVE_MAX=$lm  submission reduced:"; TBX_LIVE_MAX=$lm run --agents 128 --rollouts 128 --kv-bf16 --attn-mfma 1
done
echo "driver flags:"; ( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r05h_bench_driver_flags.log 2>&1; tail -4 $out/r05h_bench_driver_flags.log | cut -c1-300
