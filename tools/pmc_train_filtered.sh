#!/bin/bash
# HBM traffic per launch of the training step's STREAMING kernels only: the full-step PMC pass of round 5 aborted inside
# rocprofv3's counter service (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: profiles/MEASUREMENT_LOG.md); a pass restricted to a few kernels
# with --kernel-include-regex collects counters for those dispatches alone. Two passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only), each
# under its own timeout, reduced by tools/rocpd_pmc.py to gpurun_out/${TAG}_train_pmc_streaming.json.
#   TAG=r06 REGEX='tall_linear_kernel|wgrad_partial_kernel|ln_fwd_kernel|ln_bwd_kernel' bash tools/pmc_train_filtered.sh
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tag=${TAG:-r06}
regex=${REGEX:-tall_linear_kernel|wgrad_partial_kernel|ln_fwd_kernel|ln_bwd_kernel}
name=${NAME:-streaming}
out=$root/gpurun_out
cmd="bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0"
ok=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_trf_$ctr
  timeout ${PMC_TIMEOUT:-420} rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "$regex" -d $out/pmc_trf_$ctr -o p -- python3 $cmd > $out/${tag}_train_pmc_${name}_$ctr.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $ctr: exit $rc (see ${tag}_train_pmc_${name}_$ctr.log)"; ok=0; fi
done
dbs=$(find $out/pmc_trf_FETCH_SIZE $out/pmc_trf_WRITE_SIZE -name '*.db' 2>/dev/null)
if [ -n "$dbs" ]; then
  python3 tools/rocpd_pmc.py --agents 64 --polylines 1024 --lights 128 --scenes 16 --rollouts 1 --cmd "--kernel-include-regex '$regex' -- python3 $cmd" $dbs \
    > $out/${tag}_train_pmc_${name}.json 2>> $out/${tag}_train_pmc_${name}_FETCH_SIZE.log
  python3 - <<PY
import json
d = json.load(open("$out/${tag}_train_pmc_${name}.json"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:44s} launches {v['launches']:5d}  {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB per launch")
PY
fi
rm -rf $out/pmc_trf_FETCH_SIZE $out/pmc_trf_WRITE_SIZE
echo "pmc_train_filtered: ok=$ok"
