"""BASELINE.json configs[2] / [3]: the default 10M-parameter model's training_step on synthetic batches, weak scaling over ranks
(16 scenes per GPU; gradients all-reduced over RCCL when world > 1). Reference: pl_modules/waymo_motion.py:313-385, run.py:50-52."""
import os
import time
from importlib import import_module

import torch

from . import events
from .args import shard_scenes


def train_main(args, tb, dev, rank, world, dist):
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    DP = import_module("trafficbots_amd.pl_modules.data_parallel")
    # library GEMMs of the training step through rocBLAS: hipBLASLt's pick for the [n*A*W, 64] x [64, 128] input-gradient GEMMs
    # of the window PointNets runs at ~3 TF/s (105 us each, 9 per rollout step); rocBLAS: 1.03 -> 0.98 s per step (measured)
    torch.backends.cuda.preferred_blas_library(os.environ.get("TBX_BLAS", "cublas"))
    torch.manual_seed(0)  # the same initial weights on every rank ...
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    wm = wm.to(dev).train()
    wm.train_precision = getattr(args, "train_precision", None) or "bf16"  # (train_graph.py: "bf16" = autocast-class contractions, "fp32" = the fp32-class path)
    DP.broadcast_parameters(wm.model)  # ... and rank 0's by construction (one flat broadcast, as DDP's constructor does)
    (opt,), _ = wm.configure_optimizers()
    seeds = shard_scenes(args.scenes * world, rank, world)
    batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(args.scenes, args.agents, args.polylines, args.lights, seed=seeds[0]).items()}
    torch.manual_seed(DP.rank_seed(1234, rank))  # per-rank noise streams (dropout, latent, forcing)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    live = None
    if args.no_train_graph:
        state = {"live": None}
        step = lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()}, live=state["live"])
        for _ in range(args.warmup):
            step()
            state["live"] = state["live"] or DP.FlatGrads(DP.live_parameters(wm.model))  # gradients accumulate into ONE buffer from here on
        live = state["live"].params
    else:
        # forward + backward replayed as one hipGraph (the eager step is bound by the host's launch rate); the gradient
        # all-reduce, the clip and AdamW stay outside the graph. Capture (2 eager warm-up steps inside) is untimed.
        gstep = DP.GraphedTrainStep(wm, opt, batch)
        live = gstep.live
        step = lambda: gstep(batch)
        for _ in range(args.warmup):
            step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_live = sum(p.numel() for p in (live or []))
    # ---- per-kernel pass (untimed): ONE eager step with HIP events around this repo's kernels; what is not wrapped (library
    # GEMMs, aten elementwise) is the remainder of the step's GPU time, measured by an event pair around the whole step
    roof = kernels = None
    if args.profile_steps > 0:
        try:
            roof, kernels = events.train_kernel_pass(import_module("trafficbots_amd.hip"),
                                                     lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()}, live=live), dt / args.steps)
        except Exception as e:  # noqa: BLE001 - the line must still be printed
            roof = {"error": f"{type(e).__name__}: {e}"}
    return {
            "metric": "training scenes/sec", "value": world * args.scenes * args.steps / dt, "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 contractions, fp32 accumulation (autocast class); fp32 LayerNorm / softmax / attention backward / loss / AdamW"
                      if wm.train_precision == "bf16" else "f32 (split-bf16 / exact-fp32 MFMA products, VALU attention)"),
            "data": "synthetic",
            "config": {"workload": f"training_step fwd+bwd+grad all-reduce+AdamW, {args.scenes} scenes/GPU of {args.agents} agents/"
                                   f"{args.polylines} polylines/{args.lights} lights, 90-step rollout, default 10,657,094-param model",
                       "global_batch": world * args.scenes, "parallelism": f"dp{world}", "fwd_bwd_hipgraph": not args.no_train_graph,
                       "train_precision": wm.train_precision,
                       "allreduce_bytes": n_live * 4, "note": "time-batched rollout (stepping pass + one differentiated policy batch over the 90 steps); dropout as configured (p=0.1) with keyed masks: residual / FFN / MLP through tbx_keyed_dropout, "
                                                             "attention probabilities inside the HIP attention kernels"},
            "roofline": roof, "kernels": kernels,
            "loss": float(m["loss"]), "finite": bool(torch.isfinite(m["loss"]))}
