"""bench.py's command line."""
import argparse


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["rollout", "train"], default="rollout",
                    help="rollout: closed-loop sim-agent-steps/s (headline); train: training scenes/s (fwd+bwd+all-reduce+AdamW)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 80 rollout steps / 10 training steps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 10 prime steps / 3 training steps)")
    ap.add_argument("--scenes", type=int, default=None, help="scenes per GPU (default 1 rollout / 16 train)")
    ap.add_argument("--rollouts", type=int, default=1, help="parallel rollouts per scene (share the map tokens)")
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--polylines", type=int, default=1024)
    ap.add_argument("--lights", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--new-scenes", type=int, default=8, help="further scenes rolled through the same engine after the headline (end-to-end figure)")
    ap.add_argument("--repeats", type=int, default=3, help="times the timed region (W prime + K timed steps) is run; value = median")
    ap.add_argument("--pre-roll-ms", type=float, default=1500.0,
                    help="untimed device warm-up before the W warm-up steps: whole rollouts replayed and rewound for this long (0: none)")
    ap.add_argument("--graph-steps", type=int, default=40,
                    help="closed-loop steps per replayed hipGraph (the engine's own default is 4: this run replays one engine 80+ times)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--profile-steps", type=int, default=3, help="eager steps with per-kernel HIP events for the roofline")
    ap.add_argument("--no-lights-ahead", action="store_true",
                    help="one-stream engine (tl encoder -> agents -> sim step in order): the kernel-trace profiles use it so "
                         "that rocprofv3's per-kernel averages are of kernels running alone, like the live roofline events")
    ap.add_argument("--no-wosac-shape", action="store_true",
                    help="skip the second measurement (32 rollouts x 128 agents per GPU) the default rollout run appends")
    ap.add_argument("--no-train-shape", action="store_true",
                    help="skip the training_step measurement (16 scenes per GPU, fwd+bwd+all-reduce+AdamW) the default run appends")
    ap.add_argument("--no-train-graph", action="store_true", help="training: eager fwd+bwd instead of one hipGraph replay per step")
    ap.add_argument("--train-precision", choices=("bf16", "fp32"), default="bf16",
                    help="arithmetic class of the training step's contractions (train_graph.py): bf16 = autocast class (the reference trains at precision 16), "
                         "fp32 = the fp32-class parity path; the default run reports both (`training`, `training_fp32`)")
    ap.add_argument("--train-steps", type=int, default=10, help="timed training steps of that appended measurement (after 3 warm-up steps)")
    ap.add_argument("--kv-bf16", action="store_true", help="bfloat16 K/V tables (BASELINE config 2's dtype; 529 B per attention pair)")
    ap.add_argument("--linear-bf16", type=int, choices=(0, 1), default=1,
                    help="with --kv-bf16 --attn-mfma 1: Schedule.linear_bf16 (one bf16 product per LINEAR of the one-launch decoder layer; default 1 = Schedule.reduced())")
    ap.add_argument("--attn-mfma", type=int, default=None, choices=[0, 1],
                    help="Schedule.attn_mfma of the measured engines: 0 fp32 VALU attention, 1 bf16 matrix-core attention (default: the "
                         "engine's own default)")
    ap.add_argument("--no-bf16-shape", action="store_true", help="skip the bf16-table measurements the default run appends")
    ap.add_argument("--no-rule-checks", action="store_true",
                    help="skip the `with_rule_checks` figure (the timed region + the per-step rule checks of waymo_motion.py:250 + _filter_futures)")
    ap.add_argument("--no-submission-shape", action="store_true",
                    help="skip the 128 rollouts x 128 agents measurement (configs/resume/submission.yaml:5) the default run appends")
    ap.add_argument("--no-batched-shape", action="store_true",
                    help="skip the 16-scenes-per-GPU measurement (`batched`) the default run appends")
    ap.add_argument("--scene-curve", type=str, default=None, metavar="S1,S2,..",
                    help="also time the rollout at these scenes-per-GPU counts (same scene shape); one line each goes to the detail file")
    ap.add_argument("--detail-file", type=str, default=None,
                    help="where the per-kernel arrays, repeats and notes go (default gpurun_out/bench_detail.json; '-' = stderr only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo, MAX-reduce a wall time and rank 0 prints a stub line (checks the launch path)")
    a = ap.parse_args(argv)
    tr = a.mode == "train"
    a.steps = a.steps if a.steps is not None else (10 if tr else 80)   # training timed over >= 10 steps
    a.warmup = a.warmup if a.warmup is not None else (3 if tr else 10)  # after 3 warm-up steps
    # the WOSAC-shape measurement rides along only with the default (configs[1]) workload
    a.wosac_shape = (not tr and not a.no_wosac_shape and a.scenes is None and a.rollouts == 1 and a.agents == 64
                     and a.profile_steps > 0)
    a.train_shape = a.wosac_shape and not a.no_train_shape
    a.bf16_shape = a.wosac_shape and not a.no_bf16_shape and not a.kv_bf16
    a.submission_shape = a.wosac_shape and not a.no_submission_shape
    a.batched_shape = a.wosac_shape and not a.no_batched_shape
    a.rule_checks = not tr and not a.no_rule_checks and a.profile_steps > 0
    a.scenes = a.scenes if a.scenes is not None else (16 if tr else 1)
    return a


def shard_scenes(n_total: int, rank: int, world: int):
    """Scene ids simulated by `rank`: contiguous, disjoint, covering (no data-path collective is ever needed)."""
    per = (n_total + world - 1) // world
    return list(range(rank * per, min(n_total, (rank + 1) * per)))
