#!/bin/bash
# The bench shapes (configs[1], WOSAC, submission, 16 / 64 scenes, bf16 schedule) on one box with one or more builds of libtbx_hip:
# `for lib in <names>` runs csrc/libtbx_hip_<name>.so through TBX_HIP_LIB ("main" = the tree's libtbx_hip.so). Round 5 used it for
# tile_layer's plane geometry: old = the previous commit's build, ring3 / main = the new planes with TBX_TILE_RING 3 / 2
# (profiles/MEASUREMENT_LOG.md). The tile path's parity tests run first.
out=gpurun_out; mkdir -p $out
C=trafficbotsv1.5_amd/csrc
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_bf16.py -m gpu -q -x 2>&1 | tail -4 > $out/ab_tile_tests.log
run() {  # name, lib, args...
  local name=$1 lib=$2; shift 2
  TBX_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --no-rule-checks --no-submission-shape --no-batched-shape --profile-steps 0 --new-scenes 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"
}
for lib in main; do
  L=$PWD/$C/libtbx_hip_$lib.so; [ $lib = main ] && L=$PWD/$C/libtbx_hip.so
  run "c2 $lib" $L
  run "c5 $lib" $L --agents 128 --rollouts 32
  run "sub $lib" $L --agents 128 --rollouts 128 --steps 40
  run "s16 $lib" $L --scenes 16 --steps 40
  run "s64 $lib" $L --scenes 64 --steps 40
  run "c5_bf16 $lib" $L --agents 128 --rollouts 32 --kv-bf16 --attn-mfma 1
  run "s64_bf16 $lib" $L --scenes 64 --steps 40 --kv-bf16 --attn-mfma 1
done > $out/ab_tile_layer_libs.txt 2>&1
cat $out/ab_tile_tests.log $out/ab_tile_layer_libs.txt
