#!/bin/bash
# Runs on the GPU box: kernel-trace + separate PMC passes of bench.py, reduced to small summaries
# under gpurun_out/ (the rocpd databases are deleted so the pull stays small).
#   tools/profile_round.sh <tag> <agents> <polylines> <lights> <scenes> <rollouts> [more bench args...]
set -u
tag=$1; ag=$2; pl=$3; tl=$4; sc=$5; ro=$6; shift 6
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
wl="--agents $ag --polylines $pl --lights $tl --scenes $sc --rollouts $ro"
# one stream, so that a kernel's duration and counters are its own: the sequential order (--no-lights-ahead) where the timed schedule
# runs the lights on a second queue; ORDER="" for the shapes whose timed schedule already is one queue (Schedule.one_queue: configs[1])
order=${ORDER---no-lights-ahead}
kt="bench.py --no-cpu-baseline --no-wosac-shape $order $wl $*"
pmc="bench.py --no-cpu-baseline --no-wosac-shape $order --no-graph --profile-steps 0 --steps 8 --warmup 2 $wl $*"

rocprofv3 --kernel-trace --stats -d "$out/kt_$tag" -o kt -- python3 $kt > "$out/${tag}_bench.log" 2>&1
db=$(find "$out/kt_$tag" -name '*.db' | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python3 $kt"; echo; grep -h '"metric"' "$out/${tag}_bench.log" | cut -c1-400; echo;
  python3 tools/rocpd_stats.py "$db" | head -24; } > "$out/${tag}_kernel_stats.md" 2>> "$out/${tag}_bench.log"
rm -rf "$out/kt_$tag"

for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d "$out/pmc_${tag}_$ctr" -o p -- python3 $pmc > "$out/${tag}_pmc_$ctr.log" 2>&1
done
bf=""; case " $* " in *" --kv-bf16 "*) bf="--kv-bf16";; esac
python3 tools/rocpd_pmc.py $wl $bf --cmd "python3 $pmc" $(find "$out/pmc_${tag}_FETCH_SIZE" "$out/pmc_${tag}_WRITE_SIZE" -name '*.db') \
  > "$out/${tag}_pmc.json" 2>> "$out/${tag}_bench.log"
rm -rf "$out/pmc_${tag}_FETCH_SIZE" "$out/pmc_${tag}_WRITE_SIZE"
grep -h '"metric"' "$out/${tag}_bench.log" | cut -c1-200
tail -2 "$out/${tag}_bench.log" | cut -c1-200
