# A/B of the XCD-contiguous row order of the attention kernels (TBX_ATTN_XCD bit mask: 1 forward, 2 backward, 4 matrix-core forward)
F="--no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --no-rule-checks --no-submission-shape --no-batched-shape --profile-steps 0 --new-scenes 0"
run() { echo "$1 XCD=$2 $(TBX_ATTN_XCD=$2 python bench.py $F $3 2>/dev/null | tail -1 | cut -c60-175)"; }
for v in 1 0; do run "wosac fp32" $v "--agents 128 --rollouts 32 --steps 40"; done
for v in 1 0 1 0; do echo "train XCD=$v $(TBX_ATTN_XCD=$v python bench.py --mode train --steps 10 --warmup 3 --profile-steps 0 2>/dev/null | tail -1 | cut -c50-170)"; done
