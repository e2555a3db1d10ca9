# Kernel trace of the graph-replayed training step (3 replays; the last one by kernel class and phase: tools/train_replay_timeline.py), then two timed runs
root=$PWD; out=$root/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/kt_train_graph -o kt -- python3 $root/bench.py --mode train --no-cpu-baseline --steps 3 --warmup 1 --profile-steps 0 > $out/r06_train_graph.log 2>&1 )
python3 tools/train_replay_timeline.py $(ls /tmp/kt_train_graph/*.db | head -1) > $out/r06_train_replay_timeline.txt 2>&1
head -34 $out/r06_train_replay_timeline.txt
for i in 1 2; do python bench.py --mode train --steps 10 --warmup 3 --profile-steps 0 2>/dev/null | tail -1 | cut -c1-200; done
