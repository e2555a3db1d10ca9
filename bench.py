#!/usr/bin/env python
"""Headline benchmark: closed-loop sim-agent-steps/s of the HIP hot path (BASELINE.json metric, configs[1]:
synthetic 64-agent / 1024-polyline / 128-light scene, W teacher-forced prime steps (untimed warm-up) + K closed-loop
steps (timed), hipGraph replays).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes S] [--rollouts R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; scenes / rollouts are independent, so ranks never communicate on the data path (weak scaling: every rank
simulates its own scenes); the only collectives are the timing barrier and a MAX over ranks of the wall time (+ the gradient
all-reduce of the training measurement). `python bench.py --gpus N` WITHOUT a launcher starts its N ranks itself, as fresh child
processes, before anything touches the GPU (tools/benchlib/launch.py). Rank 0 prints ONE compact JSON line (< 8 KB, last line of
stdout); per-kernel arrays, repeats and notes go to the detail file (gpurun_out/bench_detail.json) and a short table to stderr.

The pieces live in tools/benchlib/: args (command line), launch (self-spawn), rollout (the timed region), training, events
(per-kernel HIP-event passes -> `roofline`), cpu (`cpu_baseline`: the oracle on the host cores), report (the line).
`value` / `ms_per_step` are the MEDIAN of `--repeats` timed regions on one engine, rewound in between. `roofline` describes the
kernel class with the largest share of the timed schedule, measured live with HIP events around every launch of a few extra
eager steps on the launch stream; `traffic` comes from the committed PMC passes (profiles/*pmc*.json, separate rocprofv3 --pmc
runs of the same workload)."""
import copy
import os
import sys
from importlib import import_module
from pathlib import Path

# ROCm 7.0 replays hipGraph memset nodes (torch's reduction semaphores) out of order on its AQL-packet fast path: a replay
# then reads sums of the previous replay (measured, tools/hipgraph_memset_repro.py). The captured training step needs the
# ordered path; the rollout graphs (kernel nodes only) run at the same speed either way. Must be set before HIP initialises.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from tools.benchlib import launch  # noqa: E402
from tools.benchlib.args import parse, shard_scenes  # noqa: E402,F401  (shard_scenes: tests/test_multiprocess_sharding.py)

DTYPE_F32 = "f32 (split-bf16 MFMA products)"  # every LINEAR of the default schedule = three bf16 MFMA products per fp32 product
DTYPE_BF16 = "bf16 (Schedule.reduced(): bf16 K/V tables + matrix-core attention operands + one bf16 product per LINEAR; fp32 accumulation, softmax, LayerNorm)"


def __getattr__(name):
    """tools/*.py import `build` / `gpu_rollout_setup` from here (they need torch: resolved lazily so that the self-spawning
    parent never imports it)."""
    if name in ("build", "gpu_rollout_setup", "scene_on_device", "engine_inputs"):
        return getattr(import_module("tools.benchlib.rollout"), name)
    raise AttributeError(name)


def dry_run(args, rank, world, report):
    """The launch path without a GPU: rendezvous (gloo), the sharding of scene ids over the ranks (gathered: disjoint and covering), the
    timing barrier + MAX-reduce of bench.py, a stub line from rank 0. --mode train additionally runs the training leg's exchange
    pattern on CPU tensors: every rank builds the default model from its OWN seed, rank 0's parameters are broadcast (checksums then
    agree on all ranks), and a FlatGrads buffer over the parameters - every gradient set to rank + 1 - goes through the one flat
    all-reduce of the training step (pl_modules/data_parallel.py), whose result must be the mean (world + 1) / 2 everywhere."""
    import time

    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo")
    t0 = time.perf_counter()
    ids = shard_scenes(args.scenes * world, rank, world)
    all_ids = [ids]
    if world > 1:
        all_ids = [None] * world
        dist.all_gather_object(all_ids, ids)
        dist.barrier()
    flat_ids = [i for r in all_ids for i in r]
    cfg = {"workload": "dry run", "scene_ids_rank0": ids, "scene_ids_per_rank": all_ids,
           "scene_ids_disjoint_and_covering": sorted(flat_ids) == list(range(args.scenes * world)) and len(set(flat_ids)) == len(flat_ids)}
    if args.mode == "train":
        from __graft_entry__ import load_package

        tb = load_package()
        DP = import_module("trafficbots_amd.pl_modules.data_parallel")
        W = import_module("trafficbots_amd.pl_modules.waymo_motion")
        torch.manual_seed(1000 + rank)  # a different initialisation on every rank ...
        wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
        sent = DP.broadcast_parameters(wm.model)  # ... made rank 0's by one flat broadcast
        chk = DP.parameters_checksum(wm.model)
        lo, hi = chk.clone(), chk.clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params = [p for p in wm.model.parameters() if p.requires_grad]
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        fg = DP.FlatGrads(params)
        nbytes = DP.allreduce_gradients(fg, world)
        want = (world + 1) / 2.0
        ok = bool(((fg.flat - want).abs() < 1e-6).all()) and all(bool(((p.grad - want).abs() < 1e-6).all()) for p in params[:4])
        seeds = [DP.rank_seed(1234, r) for r in range(world)]
        t_ok = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        cfg.update({"global_batch": world * args.scenes, "parallelism": f"dp{world}", "broadcast_bytes": sent,
                    "checksum_equal_across_ranks": bool(torch.equal(lo, hi)), "allreduce_bytes": nbytes,
                    "allreduce_is_the_mean_on_every_rank": bool(t_ok.item() == 1.0), "rank_seeds_distinct": len(set(seeds)) == world})
    t = torch.tensor([time.perf_counter() - t0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cfg["max_rank_seen"] = int(t.item())
    if rank == 0:
        report.emit({"metric": "dry-run (no GPU work)", "value": 0.0, "unit": "scenes/s" if args.mode == "train" else "sim-agent-steps/s", "n_gpus": world,
                     "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                     "dtype": DTYPE_F32, "data": "synthetic", "config": cfg, "roofline": None}, "-")
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    args = parse(argv)
    if launch.needs_spawn(args.gpus):
        # no launcher around us: start the N ranks as fresh children (nothing in this process has touched the GPU)
        sys.exit(launch.spawn_ranks(str(Path(__file__).resolve()), args.gpus, sys.argv[1:] if argv is None else argv))
    import torch

    from __graft_entry__ import load_package
    from tools.benchlib import cpu, report, rollout, training

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, rank, world, report)
    # (TBX_BENCH_SHARE_GPU=1 TBX_BENCH_BACKEND=gloo: the N-rank flow on a box with fewer GPUs than ranks - a functional check of the
    # multi-rank path, tests/test_hip_data_parallel.py; its numbers mean nothing. RCCL refuses two ranks on one device.)
    if os.environ.get("TBX_BENCH_SHARE_GPU"):
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("TBX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    hip.load()  # raises if libtbx_hip.so is missing: there is no fallback path
    if args.mode == "train":
        full = training.train_main(args, tb, dev, rank, world, dist)
        if rank == 0:
            report.emit(full, args.detail_file)
        if world > 1:
            dist.destroy_process_group()
        return

    def measure(a):
        return rollout.measure(a, tb, hip, dev, rank, world, dist)

    res, wm, full_batch = measure(args)
    full = {"metric": "sim-agent-steps/sec (closed-loop rollout)", "value": res.pop("value"), "unit": "sim-agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_BF16 if args.kv_bf16 else DTYPE_F32, "data": "synthetic", **res}
    if args.wosac_shape:
        # BASELINE.json configs[4] on the same device(s): 32 parallel rollouts x 128 agents per scenario, one scenario per
        # GPU - the size at which the relative-pose attention kernel fills the chip (its roofline fraction is the one
        # north_star's >= 50 % target refers to; the single 64-agent scene above launches 64-128 workgroups).
        big = copy.copy(args)
        big.scenes, big.rollouts, big.agents, big.steps = 1, 32, 128, min(args.steps, 40)
        r5, _, _ = measure(big)
        full["wosac_shape"] = {"metric": full["metric"], "unit": full["unit"], "n_gpus": world, "steps": big.steps, "warmup": big.warmup, **r5}
    if args.submission_shape:
        # the reference's WOSAC SUBMISSION shape (configs/resume/submission.yaml:5: 128 joint futures per scenario; 32 is the
        # training-time validation value): 128 rollouts x 128 agents = 16,384 agent rows on one GPU, the rule checks over the
        # 128 x 50 frames and the 32-of-128 filter inside `with_rule_checks`
        sub = copy.copy(args)
        sub.scenes, sub.rollouts, sub.agents, sub.steps, sub.new_scenes = 1, 128, 128, min(args.steps, 40), 0
        try:
            rs, _, _ = measure(sub)
            full["submission_shape"] = {"metric": full["metric"], "unit": full["unit"], "n_gpus": world, "steps": sub.steps, "warmup": sub.warmup, **rs}
        except Exception as e:  # noqa: BLE001
            full["submission_shape"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.batched_shape:
        # the multi-scene serving shape: 16 independent configs[1] scenes batched in one engine (the scenes-per-GPU curve's knee)
        bt = copy.copy(args)
        bt.scenes, bt.steps, bt.new_scenes = 16, min(args.steps, 40), 0
        try:
            rb, _, _ = measure(bt)
            full["batched"] = {"metric": full["metric"], "unit": full["unit"], "n_gpus": world, "steps": bt.steps, "warmup": bt.warmup, **rb}
        except Exception as e:  # noqa: BLE001
            full["batched"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.bf16_shape:
        # BASELINE.json configs[1] says bf16: the same two workloads on the bf16-arithmetic schedule (Schedule.reduced(): bfloat16 K/V
        # tables - 529 B per pair -, from 193 source rows the attention with bf16 operands on the matrix cores, and the one-launch
        # decoder layer's LINEAR stages as one bf16 product; tolerances in tests/test_hip_bf16.py / test_hip_attn_mfma.py /
        # test_hip_rollout.py). The fp32 line above stays the parity line.
        b16 = copy.copy(args)
        b16.kv_bf16, b16.attn_mfma = True, 1
        r16, _, _ = measure(b16)
        full["bf16"] = {"dtype": DTYPE_BF16, "steps": b16.steps, "warmup": b16.warmup, **r16}
        if args.wosac_shape:
            b16 = copy.copy(args)
            b16.kv_bf16, b16.attn_mfma, b16.scenes, b16.rollouts, b16.agents, b16.steps = True, 1, 1, 32, 128, min(args.steps, 40)
            r16, _, _ = measure(b16)
            full["bf16"]["wosac_shape"] = {"steps": b16.steps, "warmup": b16.warmup, **r16}
        if args.submission_shape:
            b16 = copy.copy(args)
            b16.kv_bf16, b16.attn_mfma, b16.scenes, b16.rollouts, b16.agents, b16.steps, b16.new_scenes = True, 1, 1, 128, 128, min(args.steps, 40), 0
            try:
                r16, _, _ = measure(b16)
                full["bf16"]["submission_shape"] = {"steps": b16.steps, "warmup": b16.warmup, **r16}
            except Exception as e:  # noqa: BLE001
                full["bf16"]["submission_shape"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.scene_curve:
        # the scenes-per-GPU curve (same scene shape, S scenes batched in one engine): where the one-row-per-workgroup layer
        # stops and the tile path starts. No per-kernel pass; one entry each, headline stays scenes = 1.
        full["scene_curve"] = []
        for s in [int(x) for x in args.scene_curve.split(",") if x]:
            c = copy.copy(args)
            c.scenes, c.profile_steps, c.new_scenes, c.steps = s, 0, 0, min(args.steps, 40)
            try:
                rc, _, _ = measure(c)
                full["scene_curve"].append({"scenes": s, "steps": c.steps, **{k: rc[k] for k in ("value", "ms_per_step", "ms_per_step_min", "finite")}})
            except Exception as e:  # noqa: BLE001
                full["scene_curve"].append({"scenes": s, "value": None, "ms_per_step": None, "error": f"{type(e).__name__}: {e}"[:200]})
    if args.train_shape:
        # BASELINE.json configs[2] / [3] (the metric's second half): training_step on 16 scenes per GPU, gradients
        # all-reduced over RCCL when world > 1. Every rank must take part (collective), a failure is reported, not fatal.
        tr = copy.copy(args)
        tr.scenes, tr.steps, tr.warmup, tr.agents = 16, args.train_steps, 3, 64  # >= 10 timed steps after 3 warm-ups
        try:
            full["training"] = training.train_main(tr, tb, dev, rank, world, dist)
        except Exception as e:  # noqa: BLE001
            full["training"] = {"error": f"{type(e).__name__}: {e}"}
        if tr.train_precision != "fp32":  # the fp32-class parity path beside it (no per-kernel pass)
            t32 = copy.copy(tr)
            t32.train_precision, t32.profile_steps = "fp32", 0
            try:
                full["training_fp32"] = training.train_main(t32, tb, dev, rank, world, dist)
            except Exception as e:  # noqa: BLE001
                full["training_fp32"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            full["cpu_baseline"] = cpu.cpu_baseline(tb, wm, full_batch, args)
            full["speedup_vs_cpu_baseline"] = full["value"] / full["cpu_baseline"]["value"]
        report.emit(full, args.detail_file)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
