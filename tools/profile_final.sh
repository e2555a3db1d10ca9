# Final-build evidence for a round, on the GPU box (gpurun): GPU test suite, kernel-trace + PMC passes of the 64-agent scene and the
# 32 x 128 shape with fp32 and bf16 tables (tools/profile_round.sh), SQ / cache counters of the attention kernel (tools/pmc_attn.sh),
# kernel stats of two eager training steps, and the default bench line. TAG=r04 by default; copy the summaries from gpurun_out/ into
# profiles/.
set -x
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
tag=${TAG:-r06}
if [ -z "${SKIP_TESTS:-}" ]; then timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${tag}_gpu_tests.log; fi
ORDER="" bash tools/profile_round.sh ${tag}_c2 64 1024 128 1 1   # (the one-queue step: the timed schedule itself)
bash tools/profile_round.sh ${tag}_c5 128 1024 128 1 32 --steps 40
bash tools/profile_round.sh ${tag}_c5_bf16 128 1024 128 1 32 --steps 40 --kv-bf16 --attn-mfma 1  # (Schedule.reduced(): bf16 tables + matrix-core attention)
ORDER="" bash tools/profile_round.sh ${tag}_c2_bf16 64 1024 128 1 1 --kv-bf16 --attn-mfma 1  # (Schedule.reduced() at configs[1]: bf16 tables + one bf16 product per LINEAR of the one-launch layer)
TAG=${tag}_valu bash tools/pmc_attn.sh
TAG=${tag}_mfma KPAT=%knarpe_attn_mfma_kernel% MINGRID=65536 EXTRA="--kv-bf16 --attn-mfma 1" bash tools/pmc_attn.sh
# one steady-state step of the default (one-queue) graph replay, kernel by kernel; the two-stream step beside it (TBX_ONE_QUEUE=0) and the A/B
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_final -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 --new-scenes 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_final/*.db | head -1) > gpurun_out/${tag}_c2_one_queue_timeline.txt 2>&1
rm -rf /tmp/tl_final
( cd /tmp && export TMPDIR=/tmp && TBX_ONE_QUEUE=0 rocprofv3 --kernel-trace -d /tmp/tl_final -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 --new-scenes 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_final/*.db | head -1) > gpurun_out/${tag}_c2_two_stream_timeline.txt 2>&1
rm -rf /tmp/tl_final
{ F="--no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --no-rule-checks --no-submission-shape --no-batched-shape --profile-steps 0 --new-scenes 0"; for v in 1 0 1 0; do echo "TBX_ONE_QUEUE=$v $(TBX_ONE_QUEUE=$v python bench.py $F 2>/dev/null | tail -1 | cut -c1-200)"; done; } > gpurun_out/${tag}_c2_one_queue_ab.txt
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0 > $out/${tag}_train_eager.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0  (2 eager training steps of 16 scenes; a graph replay takes the GPU-busy time of one)"; echo; tail -1 $out/${tag}_train_eager.log | cut -c1-300; echo; python3 tools/rocpd_stats.py $db 2>/dev/null | head -42; } > $out/${tag}_train_kernel_stats.md
rm -rf $out/kt_train
# phase clock inside the one-launch decoder layer (profiling build: make -C trafficbotsv1.5_amd/csrc clk)
if [ -f trafficbotsv1.5_amd/csrc/libtbx_hip_clk.so ]; then
  { echo "tools/mid_clock.py (TBX_ONE_QUEUE=0 TBX_CLOCK_TWO_STREAM=1: the two-stream schedule, eager - every half of a layer is a launch of its own): s_memtime stamps of workgroup 0 in every dec_layer_mf_kernel launch of one step (launches 0-3: the lights' 128 rows, the last with their K/V + logits tail; 4-7: the agents' 64 rows, the last with heads + tbx_sim_step + the next tbx_agent_prep; unit = 100 shader clocks, ~0.042 us)"; TBX_ONE_QUEUE=0 TBX_CLOCK_TWO_STREAM=1 python3 tools/mid_clock.py 2>/dev/null | grep -v amdgpu.ids; } > $out/${tag}_dec_layer_phase_clock.txt
fi
# round 5: the submission shape, the multi-scene shapes, the attention sweep's phase clock, the training step's HBM traffic
bash tools/profile_round.sh ${tag}_sub 128 1024 128 1 128 --steps 40 --new-scenes 0
bash tools/profile_round.sh ${tag}_s16 64 1024 128 16 1 --steps 40 --new-scenes 0
bash tools/profile_round.sh ${tag}_s64 64 1024 128 64 1 --steps 40 --new-scenes 0
bash tools/profile_round.sh ${tag}_s64_bf16 64 1024 128 64 1 --steps 40 --new-scenes 0 --kv-bf16 --attn-mfma 1
if [ -f trafficbotsv1.5_amd/csrc/libtbx_hip_clk.so ]; then
  python3 tools/attn_clock.py 2>/dev/null | grep -v amdgpu.ids > $out/${tag}_attn_phase_clock.txt
fi
# (opt-in: in round 5 the counter service aborted inside an eager training step with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT and rocprofv3
#  then sat until the job's limit - profiles/MEASUREMENT_LOG.md)
if [ -n "${TRAIN_PMC:-}" ]; then TAG=${tag} timeout 600 bash tools/pmc_train.sh > $out/${tag}_train_pmc_top.txt 2>&1; fi
# round 6: counters for the training step's streaming kernels alone (--kernel-include-regex; each pass under its own timeout), then the
# attention kernels in a pass of their own
TAG=${tag} timeout 1000 bash tools/pmc_train_filtered.sh > $out/${tag}_train_pmc_streaming_top.txt 2>&1
TAG=${tag} NAME=attention REGEX='knarpe_attn_bwd_kernel|knarpe_attn_dkv_kernel|knarpe_attn_mfma_kernel|knarpe_attn_ring_kernel' timeout 1000 bash tools/pmc_train_filtered.sh > $out/${tag}_train_pmc_attention_top.txt 2>&1
# launches of ONE graph replay of the training step (kernel trace of the timed mode: warm-up + capture + 3 replays; the last replay's window)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/kt_train_graph -o kt -- python3 $root/bench.py --mode train --no-cpu-baseline --steps 3 --warmup 1 --profile-steps 0 > $out/${tag}_train_graph.log 2>&1 )
python3 tools/train_replay_timeline.py $(ls /tmp/kt_train_graph/*.db | head -1) > $out/${tag}_train_replay_timeline.txt 2>&1
rm -rf /tmp/kt_train_graph
python3 tools/train_shape_table.py bf16 2>&1 | grep -v amdgpu.ids > $out/${tag}_train_shape_table_bf16.txt
python3 tools/train_aten_sources.py bf16 70 2>&1 | grep -v amdgpu.ids > $out/${tag}_train_aten_sources.txt
# a steady-state step of the 64-scene shape, queue by queue; the overlapped scene loop, call by call
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_s64 -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --scenes 64 --steps 40 --profile-steps 0 --new-scenes 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_s64/*.db | head -1) > $out/${tag}_s64_two_stream_timeline.txt 2>&1
rm -rf /tmp/tl_s64
timeout 300 python3 tools/scene_loop_profile.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_scene_loop_profile.txt
# the judged command, last: its line and its detail file
timeout 900 python bench.py > $out/${tag}_bench_default.log 2>&1
grep -a '"metric"' $out/${tag}_bench_default.log | tail -1 > $out/${tag}_bench_line.json
cp $out/bench_detail.json $out/${tag}_bench_detail.json
tail -1 $out/${tag}_bench_default.log | cut -c1-200
cat $out/${tag}_gpu_tests.log
