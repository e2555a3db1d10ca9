#!/usr/bin/env python
"""Headline benchmark: closed-loop sim-agent-steps/s of the HIP hot path (BASELINE.json metric, config[1]:
synthetic 64-agent / 1024-polyline / 128-light scene, 10 teacher-forced prime steps (untimed warm-up) + 80 closed-loop
steps (timed), one hipGraph replay per step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes S] [--rollouts R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; scenes / rollouts are independent, so ranks never communicate on the data path (weak scaling:
every rank simulates its own scenes); the only collectives are the timing barrier and a MAX over ranks of the wall time.
Rank 0 prints ONE JSON line. The timed region (W prime steps + K closed-loop steps) is run `--repeats` times (default 3) on the
same engine, rewound in between: `value` / `ms_per_step` are the MEDIAN repeat, the minimum rides along (`ms_per_step_min`).
`roofline` describes the kernel class with the largest share of the timed schedule (dec_layer_mf_kernel at the 64-agent scene, the row
chains / attention at the WOSAC shape; every class is listed under `kernels`), measured live with HIP events around every launch
of a few extra eager steps on the launch stream; `traffic` / `hbm_measured_frac` come from this round's committed PMC passes
(profiles/*pmc*.json, separate rocprofv3 --pmc runs); `cpu_baseline` times the oracle (CPU port of the reference formulation)
on a bounded sample of the same workload on rank 0's host cores.
"""
import argparse
import json
import os
import sys
import time
from importlib import import_module
from pathlib import Path

# ROCm 7.0 replays hipGraph memset nodes (torch's reduction semaphores) out of order on its AQL-packet fast path: a replay
# then reads sums of the previous replay (measured, tools/hipgraph_memset_repro.py). The captured training step needs the
# ordered path; the rollout graphs (kernel nodes only) run at the same speed either way. Must be set before HIP initialises.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3


def parse():
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
    ap.add_argument("--train-steps", type=int, default=10, help="timed training steps of that appended measurement (after 3 warm-up steps)")
    ap.add_argument("--kv-bf16", action="store_true", help="bfloat16 K/V tables (BASELINE config 2's dtype; 529 B per attention pair)")
    ap.add_argument("--no-bf16-shape", action="store_true", help="skip the bf16-table measurements the default run appends")
    a = ap.parse_args()
    tr = a.mode == "train"
    a.steps = a.steps if a.steps is not None else (10 if tr else 80)   # training timed over >= 10 steps
    a.warmup = a.warmup if a.warmup is not None else (3 if tr else 10)  # after 3 warm-up steps
    # the WOSAC-shape measurement rides along only with the default (configs[1]) workload
    a.wosac_shape = (not tr and not a.no_wosac_shape and a.scenes is None and a.rollouts == 1 and a.agents == 64
                     and a.profile_steps > 0)
    a.train_shape = a.wosac_shape and not a.no_train_shape
    a.bf16_shape = a.wosac_shape and not a.no_bf16_shape and not a.kv_bf16
    a.scenes = a.scenes if a.scenes is not None else (16 if tr else 1)
    return a


def shard_scenes(n_total: int, rank: int, world: int):
    """Scene ids simulated by `rank`: contiguous, disjoint, covering (no data-path collective is ever needed)."""
    per = (n_total + world - 1) // world
    return list(range(rank * per, min(n_total, (rank + 1) * per)))


def pmc_traffic(args, prefixes):
    """HBM traffic per launch from the committed PMC passes (profiles/*pmc*.json; newest round first) of this workload, for the
    kernel variant whose name starts with one of `prefixes` (the variant with the most launches in that pass)."""
    import glob

    for f in sorted(glob.glob(str(ROOT / "profiles" / "*pmc*.json")), reverse=True):
        d = json.load(open(f))
        w = d.get("workload", {})
        if (w.get("agents"), w.get("polylines"), w.get("lights"), w.get("scenes"), w.get("rollouts")) != (
                args.agents, args.polylines, args.lights, args.scenes, args.rollouts):
            continue
        if bool(w.get("kv_bf16", False)) != bool(args.kv_bf16):
            continue
        hits = [(v.get("launches", 0), k, v) for k, v in d.get("kernels", {}).items() if ("<" in k or k in prefixes) and any(k.startswith(p) for p in prefixes)]
        if hits:
            _, k, v = max(hits)
            return v["traffic_bytes_per_launch"], Path(f).name, k
    return None, None, None


L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate


def attn_counters(args):
    """VALU-busy / L2 figures of the attention kernel from the committed counter passes (profiles/*attn_counters*.json, collected by
    tools/pmc_attn.sh at the WOSAC shape) - attached only to that workload's attention entry."""
    import glob

    if (args.agents, args.rollouts, args.scenes) != (128, 32, 1):
        return None
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*attn_counters*.json")), reverse=True):
        d = json.load(open(f))
        if "valu_busy" in d and bool(d.get("kv_bf16", False)) == bool(args.kv_bf16):
            return {"valu_busy": d["valu_busy"], "l2_hit_rate": d.get("l2_hit_rate"), "l2_read_requests_per_launch": d.get("l2_read_requests"),
                    "valu_insts_per_pair": d.get("valu_insts_per_pair"), "counters_source": Path(f).name, "counters_measured": False}
    return None


def attn_algorithmic_bytes(n_src_rows: int, n_pairs: int, b: int = 4) -> float:
    """S*2*d*b + P*(2*d*b + 12 + 4 + 1) + (d_rpe*2d + 2d)*b, d = d_rpe = 128; b = 4 (fp32 tables: 1041 B per pair)
    or 2 (bfloat16 K/V tables: 529 B per pair)."""
    d = 128
    return n_src_rows * 2 * d * b + n_pairs * (2 * d * b + 17) + (d * 2 * d + 2 * d) * b


class KernelEvents:
    """Brackets every launch of the hot path's kernel classes with HIP events on the launch stream and keeps, per class, the
    algorithmic bytes (HBM-bound classes) or flops (MFMA-bound classes) of each launch:
      dec_layer  tbx_knarpe_dec_mid / tbx_knarpe_dec_layer (dec_layer_mf_kernel / dec_mid_kernel: a whole decoder layer, or its attention half)
      attn       tbx_knarpe_attn_* (knarpe_attn_kernel), grouped by source rows
      chain      tbx_rowchain / tbx_rowchain_ex (rowchain_kernel<MT,..>: MFMA row chains), grouped by tile rows
      chain_live tbx_rowchain_live (rowchain_kernel<0,1,0,1>: thread-per-column chains of small launches)
      other      K-nearest searches, preparation, tbx_sim_step (elementwise / latency)"""

    def __init__(self, hip):
        self.hip, self.rec = hip, {}
        self._saved = {}

    def _time(self, cls, key, work, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.rec.setdefault((cls, key), []).append((e0, e1, work))
        return r

    def __enter__(self):
        hip, T = self.hip, self._time
        sv = self._saved = {n: getattr(hip, n) for n in ("knarpe_attn", "knarpe_dec_mid", "knn_embed", "knn_embed_multi", "agent_prep",
                                                         "tl_prep", "sim_step", "pose_embed", "layer_tile", "heads_tile", "window_tile", "front")}
        sv["Chain.run"] = hip.Chain.run

        def mid(*args, **kw):
            # algorithmic bytes of the launch: both attentions' pairs + every weight image once (5 of the attention half;
            # with a tail the layer's out_proj / FFN / next projections = 13 chunks, with the heads 15 more) + token rows in and out
            self_seg, cross = args[4], args[5]
            rows = args[9] * args[10]
            eb = 2 if self_seg.kv.dtype == torch.bfloat16 else 4
            pairs = rows * (self_seg.k + sum(c.k for c in cross))
            tail = kw.get("tail")
            w = (4 * 33 + 36) * 2048
            if tail is not None:
                w += (8 * 33 + 3 * 32 + (3 * 33 + 36 if tail.get("qkv_out") is not None else 0) + (13 * 33 + 2 * 32 if tail.get("heads") else 0)) * 2048
            b = 2 * attn_algorithmic_bytes(rows, 0, eb) + pairs * (2 * 128 * eb + 17) + w + rows * (128 * 4 * 2 + (896 * 4 if tail and tail.get("qkv_out") is not None else 0))
            return T("dec_layer", rows, b, sv["knarpe_dec_mid"], *args, **kw)

        def attn(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, out, flag, *freqs, **kw):
            eb = 2 if segs[0].kv.dtype == torch.bfloat16 else 4
            b = attn_algorithmic_bytes(n_batch * n_src, n_batch * n_src * sum(s.k for s in segs), eb)
            return T("attn", n_batch * n_src, b, sv["knarpe_attn"], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, out, flag, *freqs, **kw)

        def run(ch, n_rows, group_rows=0):
            fl = sum(2.0 * n_rows * s.k * s.n * max(1, s.reserved) for s in ch.stages if s.op == hip.OP_LINEAR)
            if ch.live_rows:
                return T("chain_live", 0, fl, sv["Chain.run"], ch, n_rows, group_rows)
            return T("chain", ch.tile_rows, fl, sv["Chain.run"], ch, n_rows, group_rows)

        def lt(x, attn=None, ffn=None, proj=None, store_x=True, drop=None, rider=None):
            rows = x.shape[0]
            mac = (2 * 128 * 128 if attn is not None else 0) + (2 * 128 * 512 if ffn is not None else 0)
            if proj is not None:
                mac += 128 * proj["n"] + 128 * 128
            fl = 2.0 * rows * mac + (0.0 if rider is None else 2.0 * rider["out"].shape[0] * 4 * 128 * 128)
            return T("tile", "layer", fl, sv["layer_tile"], x, attn=attn, ffn=ffn, proj=proj, store_x=store_x, drop=drop, rider=rider)

        def ht(x, hd):
            return T("tile", "heads", 2.0 * x.shape[0] * (2 * (256 * 128 + 2 * 128 * 128) + 128 * 384 + 3 * 128 * 128 + 3 * 128 * 16), sv["heads_tile"], x, hd)

        def wt(attr, pe, row_invalid, in_images, pn_images, window, out, add_mode=False, drop=None):
            mac = (32 * 128 + 2 * 128 * 128 if add_mode else 32 * 64 + 2 * 64 * 64) + 3 * 128 * 64
            return T("tile", "window", 2.0 * attr.shape[0] * mac, sv["window_tile"], attr, pe, row_invalid, in_images, pn_images, window, out,
                     add_mode=add_mode, drop=drop)

        def fr(window, proj, rider=None, jobs=None, pose_embed_job=None):
            rows = window["out"].shape[0]
            add = bool(window.get("add_mode"))
            mac_w = (32 * 128 + 2 * 128 * 128 if add else 32 * 64 + 2 * 64 * 64) + 3 * 128 * 64
            fl = 2.0 * window["attr"].shape[0] * mac_w + 2.0 * rows * (128 * 384 + 128 * 128) + (0.0 if rider is None else 2.0 * rider["out"].shape[0] * 4 * 128 * 128)
            return T("tile", "front", fl, sv["front"], window, proj, rider=rider, jobs=jobs, pose_embed_job=pose_embed_job)

        def other(name):
            return lambda *a, **kw: T("other", name, 0.0, sv[name], *a, **kw)

        hip.knarpe_attn, hip.Chain.run, hip.knarpe_dec_mid = attn, run, mid
        hip.layer_tile, hip.heads_tile, hip.window_tile, hip.front = lt, ht, wt, fr
        for n in ("knn_embed", "knn_embed_multi", "agent_prep", "tl_prep", "sim_step", "pose_embed"):
            setattr(hip, n, other(n))
        return self

    def __exit__(self, *a):
        for n, f in self._saved.items():
            if n == "Chain.run":
                self.hip.Chain.run = f
            else:
                setattr(self.hip, n, f)

    def classes(self, n_steps: int):
        """-> list of per-(class, key) dicts sorted by total time, largest first."""
        torch.cuda.synchronize()
        out = []
        for (cls, key), evs in self.rec.items():
            t = sum(e0.elapsed_time(e1) for e0, e1, _ in evs) * 1e-3
            out.append(dict(cls=cls, key=key, t=t, n=len(evs), work=sum(w for *_, w in evs), per_step=len(evs) / n_steps))
        tot = sum(c["t"] for c in out) or 1.0
        for c in out:
            c["share"] = c["t"] / tot
        return sorted(out, key=lambda c: -c["t"])


def kernel_entry(args, c):
    """One `kernels` / `roofline` object for a KernelEvents class: achieved = algorithmic bytes (or flops) per launch / the average
    launch duration between HIP events; traffic = HBM bytes per launch from this workload's committed PMC pass (if any)."""
    cls, key = c["cls"], c["key"]
    avg = c["t"] / c["n"]
    e = {"class": cls, "share_of_step_kernel_time": c["share"], "launches_per_step": c["per_step"], "avg_launch_us": avg * 1e6}
    if cls in ("dec_layer", "attn"):
        ach = c["work"] / c["t"] / 1e9
        name = "dec_layer_mf_kernel" if cls == "dec_layer" else "knarpe_attn_kernel"  # (dec_mid_kernel with Schedule.dec_tail_mfma off)
        pre = ["dec_layer_mf_kernel<", "dec_mid_kernel<"] if cls == "dec_layer" else (["knarpe_attn_kernel<1,"] if key >= 1024 else ["knarpe_attn_kernel<4,"])
        e.update(kernel=name, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                 algorithmic_bytes_per_launch=c["work"] / c["n"], source_rows_per_launch=key,
                 bytes_per_pair=529 if args.kv_bf16 else 1041)
        if cls == "attn":
            e.update(l2_frac=ach / L2_PEAK_GBS, l2_peak=L2_PEAK_GBS)
        if cls == "dec_layer" or key < 1024:
            e["note"] = ("latency-bound at this size: a launch has one workgroup per source row (64-128 of them on 256 CUs) and the "
                         "step is a chain of dependent launches; frac is bytes over time, not a bandwidth-limited figure")
    elif cls in ("chain", "chain_live", "tile"):
        ach = c["work"] / c["t"] / 1e12
        if cls == "tile":
            name = "front_kernel" if key == "front" else f"tile_{key}_kernel"  # (tbx_front: window tile + first projection + searches)
            pre = [name]
        else:
            name = "rowchain_kernel" + ("<live>" if cls == "chain_live" else f"<{key}-row tiles>")
            pre = ["rowchain_kernel<0,1,0,1>"] if cls == "chain_live" else [f"rowchain_kernel<{key // 16},"]
        e.update(kernel=name, bound="mfma", achieved=ach, peak=FP32_MFMA_PEAK_TF,
                 unit="TFLOP/s", frac=ach / FP32_MFMA_PEAK_TF, flops_per_launch=c["work"] / c["n"],
                 peak_note="dense fp32 MFMA peak: the work is fp32 LINEAR stages (exact-fp32 MFMA in the row chains; the tile kernels form "
                           "each fp32 product from three bf16 MFMA products, priced against the same fp32 peak, not the bf16 one)")
    else:
        e.update(kernel=f"tbx_{key}", bound="latency", achieved=None, peak=None, unit=None, frac=None)
        return e
    traffic, src, variant = pmc_traffic(args, pre)
    e.update(traffic=traffic, traffic_source=src, traffic_kernel=variant,
             traffic_measured=False)  # PMC passes are separate rocprofv3 runs: the committed profile of this workload
    if traffic is not None:
        e["hbm_measured_frac"] = traffic / avg / 1e9 / HBM_PEAK_GBS
    return e


def build(tb, args, dev, rank):
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    torch.manual_seed(0)
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    wm = wm.to(dev).eval()  # random init of the reference architecture (no checkpoint on the box)
    # weak scaling: rank r simulates scenes [r*S, (r+1)*S) of the global list (seed = scene id)
    seeds = shard_scenes(args.scenes * int(os.environ.get("WORLD_SIZE", 1)), rank, int(os.environ.get("WORLD_SIZE", 1)))
    batch = tb.synthetic.make_scene(args.scenes, args.agents, args.polylines, args.lights, seed=seeds[0])
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    return wm, full


def scene_on_device(tb, wm, args, dev, seed):
    """A synthetic scene batch of this workload's shape (seed = scene id), pre-processed, resident in HBM."""
    batch = tb.synthetic.make_scene(args.scenes, args.agents, args.polylines, args.lights, seed=seed)
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    return wm.pre_processing({k: v.to(dev) for k, v in full.items()})


def engine_inputs(wm, bd, args, dev, n_step):
    """Once-per-scene work (map encoder, traffic-light pre-compute, K/V tables) + the arguments of RolloutEngine.reset / refill."""
    R = args.rollouts
    mp, tl = wm.encode_scene(bd, n_rollout=R)
    r = (lambda t: t.repeat_interleave(R, 0)) if R > 1 else (lambda t: t)
    n, A = args.scenes * R, args.agents
    g = torch.Generator().manual_seed(0)
    z = torch.randn(n, A, 16, generator=g).to(dev)  # prior sample (std-normal), injected
    valid = r(bd["sc/ag_valid"].any(-1))
    tf = wm.teacher_forcing_joint_future_pred
    tf.init(ag_valid=r(bd["sc/ag_valid"]), ag_pose=r(bd["sc/ag_pose"]), ag_motion=r(bd["sc/ag_motion"]),
            tl_state=r(bd["sc/tl_state"]), current_epoch=0)
    return dict(gt_valid=r(bd["sc/ag_valid"]), gt_pose=r(bd["sc/ag_pose"]), gt_motion=r(bd["sc/ag_motion"]),
                tl_state_gt=r(bd["sc/tl_state"]), tf_mask=tf.ag_teacher_forcing, ag_type=r(bd["ref/ag_type"]),
                ag_attr=r(bd["sc/ag_attr"]), ag_latent=z, ag_latent_valid=valid, ag_navi=r(bd["gt/ag_navi"]), ag_navi_valid=valid,
                mp_tokens=mp, tl_tokens=tl, map_valid=bd["map/valid"], map_type=bd["map/type"], map_pos=bd["map/pos"],
                map_dir=bd["map/dir"], map_boundary=bd["map/boundary"], n_step=n_step)


def gpu_rollout_setup(tb, wm, full, args, dev):
    bd = wm.pre_processing({k: v.to(dev) for k, v in full.items()})
    t0 = time.perf_counter()
    kw = engine_inputs(wm, bd, args, dev, args.warmup + args.steps + 2 * args.profile_steps)
    torch.cuda.synchronize()
    t_scene = time.perf_counter() - t0
    Eng = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine
    eng = Eng(wm.model, wm.dynamics, dev, schedule=wm.schedule)
    eng.reset(**kw)
    return eng, t_scene


def cpu_baseline(tb, wm, full, args):
    """Oracle (CPU port, reference formulation) on a bounded sample: first scene, `cpu_steps` closed-loop steps."""
    from oracle import trafficbots_oracle as O

    P = {k: v.detach().cpu().clone() for k, v in wm.model.state_dict().items()}
    one = {k: v[:1] for k, v in full.items()}
    b = O.scene_centric(one, training=False)
    cfg, scfg = tb.config.default_model_cfg(), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, args.agents, 16, generator=g)
    valid = b["sc/ag_valid"].any(-1)
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    with torch.no_grad():
        mp = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp)
        sim = O.Sim(om, scfg, False)
        run = lambda n: sim.rollout(bh, mp, tl, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, n,
                                    gt_prefix="hist", tl_gt_key="sc/tl_state")
        run(2)  # warm up thread pools / allocator
        # pick the thread count that is fastest for this small-op workload (all cores is rarely it), then time
        best, n_all = None, torch.get_num_threads()
        for nt in sorted({8, 16, 32, 64, n_all} & set(range(1, n_all + 1))):
            torch.set_num_threads(nt)
            run(2)
            t0 = time.perf_counter()
            run(4)
            d = time.perf_counter() - t0
            if best is None or d < best[0]:
                best = (d, nt)
        torch.set_num_threads(best[1])
        # bounded sample: whole `cpu_steps`-step rollouts until ~10 s of CPU work (the host cores of a box are shared and
        # their speed varies an order of magnitude between boxes; a sub-second sample is noise)
        n_done, t0 = 0, time.perf_counter()
        while n_done == 0 or (time.perf_counter() - t0 < 10.0 and n_done < 40 * args.cpu_steps):
            run(args.cpu_steps)
            n_done += args.cpu_steps
        dt = time.perf_counter() - t0
        torch.set_num_threads(n_all)
    return {"value": args.agents * n_done / dt, "unit": "sim-agent-steps/s", "cores": best[1],
            "kind": "port", "sample": f"1 scene x {args.agents} agents x {n_done} closed-loop steps ({n_done // args.cpu_steps} rollouts of {args.cpu_steps}) in {dt:.1f}s "
                                      f"(oracle, torch {torch.__version__} CPU fp32, best of 8/16/32/64/all = {best[1]} threads of {n_all}, "
                                      f"map encoding excluded)"}


def train_kernel_pass(hip, step, replay_s):
    """Times this repo's kernels inside one eager training step (HIP events on the launch stream, behind a device-side delay that lets
    the host enqueue the step ahead of the device: the pairs then bracket back-to-back launches; shares are of `replay_s`, the timed
    hipGraph replay of the same launches). Algorithmic work:
    attention forward = SURVEY 8d bytes; backward = the forward's bytes + d(out) and d(q) rows (1280 floats per row) + 8 coefficient
    floats per pair; tbx_linear_wgrad = dY and X read once (4 (n + k) bytes per row); tbx_tall_linear = X read, Y written once (the
    same 4 (n + k) bytes per row); LayerNorm 1.0 / 1.5 KB per row; chains / tile kernels: flops."""
    rec, saved = {}, {}

    def T(cls, bound, work, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        rec.setdefault((cls, bound), []).append((e0, e1, work))
        return r

    def pairs(n_batch, n_src, segs):
        return n_batch * n_src, n_batch * n_src * sum(sg.k for sg in segs)

    def attn(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw):
        r, p = pairs(n_batch, n_src, segs)
        return T("knarpe_attn_kernel (forward)", "hbm", attn_algorithmic_bytes(r, p), saved["knarpe_attn"], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw)

    def attn_bwd(name):
        def f(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw):
            r, p = pairs(n_batch, n_src, segs)
            return T("knarpe_attn_bwd_kernel + dkv", "hbm", attn_algorithmic_bytes(r, p) + r * 1280 * 4 + p * 32, saved[name], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw)
        return f

    def wgrad(dy, x, *a, **kw):
        return T("wgrad_partial_kernel (tbx_linear_wgrad)", "hbm", 4.0 * dy.shape[0] * (dy.shape[1] + x.shape[1]), saved["linear_wgrad"], dy, x, *a, **kw)

    def tall(x, w, b=None, wt=False, relu=False):
        n_, k_ = (w.shape[1], w.shape[0]) if wt else (w.shape[0], w.shape[1])
        return T("tall_linear_kernel (tbx_tall_linear)", "hbm", 4.0 * (x.numel() // k_) * (k_ + n_), saved["tall_linear"], x, w, b, wt=wt, relu=relu)

    def lt(x, attn=None, ffn=None, proj=None, store_x=True, drop=None, rider=None):
        mac = (2 * 128 * 128 if attn is not None else 0) + (2 * 128 * 512 if ffn is not None else 0) + (0 if proj is None else 128 * proj["n"] + 128 * 128)
        return T("tile_layer / tile_heads / tile_window kernels (stepping pass)", "mfma", 2.0 * x.shape[0] * mac, saved["layer_tile"], x, attn=attn, ffn=ffn,
                 proj=proj, store_x=store_x, drop=drop, rider=rider)

    def ln_f(x, *a, **kw):
        return T("ln_fwd_kernel", "hbm", x.numel() * 8.0, saved["layernorm_fwd"], x, *a, **kw)

    def ln_b(x, *a, **kw):
        return T("ln_bwd_kernel", "hbm", x.numel() * 12.0, saved["layernorm_bwd"], x, *a, **kw)

    def run(ch, n_rows, group_rows=0):
        fl = sum(2.0 * n_rows * st.k * st.n * max(1, st.reserved) for st in ch.stages if st.op == hip.OP_LINEAR)
        return T("rowchain_kernel (stepping pass)", "mfma", fl, saved["Chain.run"], ch, n_rows, group_rows)

    names = {"knarpe_attn": attn, "knarpe_attn_bwd_gather": attn_bwd("knarpe_attn_bwd_gather"), "knarpe_attn_bwd": attn_bwd("knarpe_attn_bwd"),
             "linear_wgrad": wgrad, "layernorm_fwd": ln_f, "layernorm_bwd": ln_b, "tall_linear": tall, "layer_tile": lt}
    for n, f in names.items():
        saved[n] = getattr(hip, n)
        setattr(hip, n, f)
    saved["Chain.run"] = hip.Chain.run
    hip.Chain.run = run
    try:
        # the eager step is bound by the host's launch rate: on an idle stream an event pair around a launch times the wait for the host
        # to enqueue it (seen here: 515 us "launches" of a 73 us kernel). A device-side delay in front, as long as the host needs to
        # enqueue the whole step (measured on one plain eager step first), lets the launches queue up and run back to back.
        torch.cuda.synchronize()
        t_host = time.perf_counter()
        saved_step = {n: getattr(hip, n) for n in names}
        for n in names:  # (the plain step: unwrapped)
            setattr(hip, n, saved[n])
        hip.Chain.run = saved["Chain.run"]
        step()
        t_host = time.perf_counter() - t_host  # (enqueue time: nothing in the step waits for the device)
        torch.cuda.synchronize()
        for n, f in saved_step.items():
            setattr(hip, n, f)
        hip.Chain.run = run
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(2_000_000)  # (calibration: what one spin cycle of torch's delay kernel is on this device)
        c1.record()
        torch.cuda.synchronize()
        cycles_per_s = 2e6 / max(1e-6, c0.elapsed_time(c1) * 1e-3)
        torch.cuda._sleep(int(cycles_per_s * min(3.0, 1.5 * t_host + 0.05)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
    finally:
        for n, f in saved.items():
            if n == "Chain.run":
                hip.Chain.run = f
            else:
                setattr(hip, n, f)
    total = e0.elapsed_time(e1) * 1e-3
    kernels = []
    for (cls, bound), evs in rec.items():
        t = sum(a.elapsed_time(b) for a, b, _ in evs) * 1e-3
        w = sum(x for *_, x in evs)
        peak, unit, ach = (HBM_PEAK_GBS, "GB/s", w / t / 1e9) if bound == "hbm" else (FP32_MFMA_PEAK_TF, "TFLOP/s", w / t / 1e12)
        kernels.append({"kernel": cls, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "launches_per_step": len(evs),
                        "avg_launch_us": t / len(evs) * 1e6, "share_of_step": t / replay_s, "traffic": None})
    kernels.sort(key=lambda k: -k["share_of_step"])
    rest = 1.0 - sum(k["share_of_step"] for k in kernels)
    kernels.append({"kernel": "library GEMMs of the odd-width layers (rocBLAS fp32) + aten elementwise / copy / reduce + this repo's smaller kernels", "bound": None,
                    "share_of_step": rest})
    roof = dict(kernels[0])
    roof["note"] = ("largest of this repo's kernel classes in ONE eager training step enqueued behind a device-side delay (event pairs on the "
                    "launch stream around back-to-back launches); share_of_step = its event time over the timed hipGraph replay of the same launches")
    roof["eager_step_device_ms"] = total * 1e3
    roof["eager_step_enqueue_ms"] = t_host * 1e3
    return roof, kernels


def train_main(args, tb, dev, rank, world, dist):
    """Config 3/4: default 10M-parameter model, training_step on synthetic batches, weak scaling over ranks."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    DP = import_module("trafficbots_amd.pl_modules.data_parallel")
    # library GEMMs of the training step through rocBLAS: hipBLASLt's pick for the [n*A*W, 64] x [64, 128] input-gradient GEMMs
    # of the window PointNets runs at ~3 TF/s (105 us each, 9 per rollout step); rocBLAS: 1.03 -> 0.98 s per step (measured)
    torch.backends.cuda.preferred_blas_library(os.environ.get("TBX_BLAS", "cublas"))
    torch.manual_seed(0)  # the same initial weights on every rank ...
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    wm = wm.to(dev).train()
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
            roof, kernels = train_kernel_pass(import_module("trafficbots_amd.hip"), lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()}, live=live),
                                              dt / args.steps)
        except Exception as e:  # noqa: BLE001 - the line must still be printed
            roof = {"error": f"{type(e).__name__}: {e}"}
    return {
            "metric": "training scenes/sec", "value": world * args.scenes * args.steps / dt, "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"training_step fwd+bwd+grad all-reduce+AdamW, {args.scenes} scenes/GPU of {args.agents} agents/"
                                   f"{args.polylines} polylines/{args.lights} lights, 90-step rollout, default 10,657,094-param model",
                       "global_batch": world * args.scenes, "parallelism": f"dp{world}", "fwd_bwd_hipgraph": not args.no_train_graph,
                       "allreduce_bytes": n_live * 4, "note": "time-batched rollout (stepping pass + one differentiated policy batch over the 90 steps); dropout as configured (p=0.1) with keyed masks: residual / FFN / MLP through tbx_keyed_dropout, "
                                                             "attention probabilities inside the HIP attention kernels"},
            "roofline": roof, "kernels": kernels,
            "loss": float(m["loss"]), "finite": bool(torch.isfinite(m["loss"]))}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    hip.load()
    if args.mode == "train":
        line = train_main(args, tb, dev, rank, world, dist if world > 1 else None)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    E = import_module("trafficbots_amd.engine")

    def measure(a):
        """The timed closed-loop rollout of workload `a` on this rank (a.repeats times); returns the JSON fields of that measurement."""
        wm, full = build(tb, a, dev, rank)
        # (a timed region shorter than --graph-steps: one graph of all of it; after an odd number of warm-up steps the light tables'
        # double buffer is at parity 1 and the multi-step graph, captured at parity 0, starts one step in)
        gsteps = max(1, min(a.graph_steps, a.steps - (a.warmup % 2)) // 2 * 2)
        # this measurement's schedule belongs to its module / engine (engine.Schedule), not to the process
        wm.schedule = E.DEFAULT.replace(kv_bf16=bool(a.kv_bf16), lights_ahead=not a.no_lights_ahead, graph_steps=gsteps)
        eng, t_scene = gpu_rollout_setup(tb, wm, full, a, dev)
        use_graph = not a.no_graph
        t_cap = time.perf_counter()
        if use_graph:
            eng.capture()
            torch.cuda.synchronize()
        t_cap = time.perf_counter() - t_cap
        # device pre-roll (untimed, not part of W): the timed region is ~25 ms of a chain of 20-40 us launches, and a device that
        # was idle a moment ago runs its first hundreds of milliseconds below its steady clocks (the same binary measured 193 k,
        # 195 k, 200 k agent-steps/s in three consecutive processes). Whole rollouts are replayed and rewound until
        # --pre-roll-ms of wall time have passed.
        t_pre, n_pre = time.perf_counter(), 0
        while use_graph and (time.perf_counter() - t_pre) * 1e3 < a.pre_roll_ms:
            eng.run(a.warmup + a.steps, use_graph=True)
            torch.cuda.synchronize()
            eng.restore()
            n_pre += 1
        dts = []
        for rep in range(max(1, a.repeats)):  # the region is timed >= 3 times; median and minimum are reported
            if rep:
                eng.restore()
            eng.run(a.warmup, use_graph=use_graph)  # teacher-forced prime steps (untimed)
            barrier()
            t0 = time.perf_counter()
            eng.run(a.steps, use_graph=use_graph)
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            dts.append(dt)
        dt = sorted(dts)[len(dts) // 2]
        units = world * a.scenes * a.rollouts * a.agents * a.steps
        timing = {"value": units / dt, "ms_per_step": dt / a.steps * 1e3, "repeats": len(dts), "ms_per_step_min": min(dts) / a.steps * 1e3,
                  "ms_per_step_all": [d / a.steps * 1e3 for d in dts], "value_best": units / min(dts)}
        # ---- scene-to-scene reuse (the reference's validation_step loops over scenes, waymo_motion.py:526): NEW scenes through the
        # same engine - once-per-scene encoders + RolloutEngine.refill (in place: the captured graphs stay valid) + the W prime and K
        # closed-loop steps, everything timed; scene tensors resident in HBM as in the headline. No graph capture in this loop.
        reuse = None
        if use_graph and a.new_scenes > 0 and world == 1:
            first = shard_scenes(a.scenes * world, rank, world)[0]
            bds = [scene_on_device(tb, wm, a, dev, first + 1000 + i) for i in range(a.new_scenes)]
            with E.use(wm.schedule):
                # two untimed scenes first (refill + rollout): the first refills of a process pay one-time costs (allocator growth;
                # measured 90 ms, once, in the first OR the second refill) - the figure is the steady state of a loop over scenes
                for w_ in range(2):
                    eng.refill(**engine_inputs(wm, scene_on_device(tb, wm, a, dev, first + 998 + w_), a, dev, a.warmup + a.steps + 2 * a.profile_steps))
                    eng.run(a.warmup + a.steps, use_graph=True)
            torch.cuda.synchronize()
            t_enc, t_all = [], time.perf_counter()
            with E.use(wm.schedule):
                for bd in bds:
                    torch.cuda.synchronize()  # (attribution only: the previous scene's rollout is done before this one's clock starts)
                    t0 = time.perf_counter()
                    eng.refill(**engine_inputs(wm, bd, a, dev, a.warmup + a.steps + 2 * a.profile_steps))
                    torch.cuda.synchronize()
                    t_enc.append(time.perf_counter() - t0)
                    eng.run(a.warmup + a.steps, use_graph=True)
                torch.cuda.synchronize()
            t_all = time.perf_counter() - t_all
            reuse = {"scenes": a.new_scenes, "new_scene_ms": sorted(t_enc)[len(t_enc) // 2] * 1e3, "new_scene_ms_all": [t * 1e3 for t in t_enc],
                     "end_to_end_value": a.new_scenes * a.scenes * a.rollouts * a.agents * a.steps / t_all,
                     "ms_per_scene": t_all / a.new_scenes * 1e3,
                     "note": "per new scene: map encoder + light pre-compute + K/V tables + RolloutEngine.refill (new_scene_ms), then W prime + K "
                             "closed-loop steps on the graphs captured once for this shape; end_to_end_value counts the K steps' agent-steps "
                             "over ALL of that time"}
            eng.restore()
        if a.profile_steps <= 0:  # tooling only (timeline traces, A/B runs): the timed region without the per-kernel pass
            return {**timing, "roofline": None, "cpu_baseline": None,
                    "note": "--profile-steps 0: no per-kernel timing pass, not a judged line",
                    "finite": bool(torch.isfinite(eng.S["out_pose"]).all())}, wm, full
        # ---- live per-kernel timing: eager steps right after the timed region, same state and the SAME launches as the timed
        # schedule, events on the launch stream, in the engine's one-stream order so that a kernel's duration is its own (in the
        # timed region the light and agent halves share the device, which stretches the kernels of both)
        eng.sched = eng.sched.replace(lights_ahead=False)
        # the host must be AHEAD of the device while the events are recorded: an event pair around a launch otherwise also
        # times the wait for the host to enqueue that launch (seen on a loaded box: 27 us "launches" of a 10 us kernel).
        # A device-side delay in front lets the host queue all launches of the profiled steps first.
        torch.cuda._sleep(int(2.4e9 * (0.01 + 0.006 * a.profile_steps)))
        with KernelEvents(hip) as ke:
            eng.run(a.profile_steps, use_graph=False)
        classes = ke.classes(a.profile_steps)
        kernels = [kernel_entry(a, c) for c in classes]
        # the judged object: the kernel class the largest share of the step's kernel time goes to
        roof = dict(next(k for k in kernels if k["bound"] != "latency"))
        cnt = attn_counters(a)
        att = next((k for k in kernels if k["class"] == "attn" and k["source_rows_per_launch"] >= 1024), None)
        if cnt and att is not None:
            if cnt.get("l2_read_requests_per_launch"):  # 128-byte L1 -> L2 read requests of a launch over its live duration
                cnt["l2_request_frac"] = cnt["l2_read_requests_per_launch"] * 128.0 / (att["avg_launch_us"] * 1e-6) / 1e9 / L2_PEAK_GBS
            att["counters"] = cnt
        mfma = [k for k in kernels if k["bound"] == "mfma"]
        res = {
            **timing,
            "config": {"workload": f"{a.agents}-agent/{a.polylines}-polyline/{a.lights}-light synthetic scene, "
                                   f"{a.warmup}-step teacher-forced prime + {a.steps}-step closed-loop rollout",
                       "scenes_per_gpu": a.scenes, "rollouts_per_scene": a.rollouts, "graph": use_graph,
                       "steps_per_graph_replay": gsteps if use_graph else 0,
                       "pre_roll_rollouts": n_pre,  # untimed whole-rollout replays before the W warm-up steps (device at steady clocks)
                       "lights_one_step_ahead_on_second_stream": not a.no_lights_ahead,
                       "weights": "random init of the 10,657,094-parameter default architecture"},
            "roofline": roof,
            "kernels": kernels,  # every kernel class of the step, largest share first (roofline = the first non-elementwise one)
            "roofline_gemm": None if not mfma else {"kernel": "rowchain_kernel (all row chains)", "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MFMA_PEAK_TF,
                                                    "achieved": sum(k["flops_per_launch"] * k["launches_per_step"] for k in mfma) /
                                                                sum(k["avg_launch_us"] * 1e-6 * k["launches_per_step"] for k in mfma) / 1e12},
            "scene_encode_ms": t_scene * 1e3, "graph_capture_ms": t_cap * 1e3,
            # SURVEY §8d "end-to-end": the once-per-scene work (map encoder, traffic-light pre-compute, K/V tables, engine refill)
            # counted into the same units, over new scenes rolled through the SAME engine; the first scene of a process additionally
            # pays scene_encode_ms (cold: allocations, weight packing) and graph_capture_ms once per shape
            "end_to_end_value": reuse["end_to_end_value"] if reuse else units / (dt + t_scene),
            "scene_reuse": reuse,
            "first_scene_value": units / (dt + t_scene + t_cap),
            "finite": bool(torch.isfinite(eng.S["out_pose"]).all()),
        }
        if res["roofline_gemm"]:
            res["roofline_gemm"]["frac"] = res["roofline_gemm"]["achieved"] / FP32_MFMA_PEAK_TF
        return res, wm, full

    res, wm, full = measure(args)
    line = {"metric": "sim-agent-steps/sec (closed-loop rollout)", "value": res.pop("value"), "unit": "sim-agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 K/V tables, f32 arithmetic" if args.kv_bf16 else "f32", "data": "synthetic", **res}
    if args.wosac_shape:
        # BASELINE.json configs[4] on the same device(s): 32 parallel rollouts x 128 agents per scenario, one scenario per
        # GPU - the size at which the relative-pose attention kernel fills the chip (its roofline fraction is the one
        # north_star's >= 50 % target refers to; the single 64-agent scene above launches 64-128 workgroups).
        import copy

        big = copy.copy(args)
        big.scenes, big.rollouts, big.agents, big.steps = 1, 32, 128, min(args.steps, 40)
        r5, _, _ = measure(big)
        line["wosac_shape"] = {"metric": line["metric"], "unit": line["unit"], "n_gpus": world, "steps": big.steps,
                               "warmup": big.warmup, **r5}
    if args.bf16_shape:
        # BASELINE.json configs[1] says bf16: the same two workloads with bfloat16 K/V tables (engine.KV_BF16: 529 B per pair,
        # fp32 queries / embeddings / softmax / sums; tolerances in tests/test_hip_bf16.py). The fp32 line above stays the parity line.
        import copy

        b16 = copy.copy(args)
        b16.kv_bf16 = True
        r16, _, _ = measure(b16)
        line["bf16"] = {"dtype": "bf16 K/V tables, f32 arithmetic", "steps": b16.steps, "warmup": b16.warmup, **r16}
        if args.wosac_shape:
            b16 = copy.copy(args)
            b16.kv_bf16, b16.scenes, b16.rollouts, b16.agents, b16.steps = True, 1, 32, 128, min(args.steps, 40)
            r16, _, _ = measure(b16)
            line["bf16"]["wosac_shape"] = {"steps": b16.steps, "warmup": b16.warmup, **r16}
    if args.train_shape:
        # BASELINE.json configs[2] / [3] (the metric's second half): training_step on 16 scenes per GPU, gradients
        # all-reduced over RCCL when world > 1. Every rank must take part (collective), a failure is reported, not fatal.
        import copy

        tr = copy.copy(args)
        tr.scenes, tr.steps, tr.warmup, tr.agents = 16, args.train_steps, 3, 64  # >= 10 timed steps after 3 warm-ups
        try:
            line["training"] = train_main(tr, tb, dev, rank, world, dist if world > 1 else None)
        except Exception as e:  # noqa: BLE001
            line["training"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(tb, wm, full, args)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
