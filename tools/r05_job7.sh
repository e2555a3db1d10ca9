#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
( time python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) > $out/r05_gpu_tests_full.log 2>&1
tail -8 $out/r05_gpu_tests_full.log
( time python bench.py ) > $out/r05f_bench_default.log 2>&1
grep -a '"metric"' $out/r05f_bench_default.log | tail -1 > $out/r05f_bench_line.json
wc -c $out/r05f_bench_line.json; tail -4 $out/r05f_bench_default.log | cut -c1-300
