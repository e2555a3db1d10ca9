#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
timeout 300 python3 tools/attn_clock.py > $out/r05_attn_phase_clock.txt 2> $out/r05_attn_phase_clock.err; tail -3 $out/r05_attn_phase_clock.err | cut -c1-300; head -30 $out/r05_attn_phase_clock.txt | cut -c1-200
timeout 300 python3 tools/scene_loop_profile.py > $out/r05_scene_loop_profile.txt 2>&1; tail -8 $out/r05_scene_loop_profile.txt | cut -c1-200
( time timeout 600 python bench.py ) > $out/r05_bench_default.log 2>&1
grep -a '"metric"' $out/r05_bench_default.log | tail -1 > $out/r05_bench_line.json; wc -c $out/r05_bench_line.json
cp $out/bench_detail.json $out/r05_bench_detail.json
tail -3 $out/r05_bench_default.log | cut -c1-100
