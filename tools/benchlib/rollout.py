"""The timed closed-loop rollout of one workload on this rank: engine set-up, graph capture, device pre-roll, `--repeats` timed
regions (W teacher-forced prime steps untimed + K closed-loop steps timed), new scenes through the same engine (end-to-end figure),
and the per-kernel-class HIP-event pass the `roofline` object comes from. Reference loop: pl_modules/waymo_motion.py:118-204."""
import os
import time
from importlib import import_module

import torch

from . import events
from .args import shard_scenes


def build(tb, args, dev, rank):
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    torch.manual_seed(0)
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    wm = wm.to(dev).eval()  # random init of the reference architecture (no checkpoint on the box)
    # weak scaling: rank r simulates scenes [r*S, (r+1)*S) of the global list (seed = scene id)
    world = int(os.environ.get("WORLD_SIZE", 1))
    seeds = shard_scenes(args.scenes * world, rank, world)
    batch = tb.synthetic.make_scene(args.scenes, args.agents, args.polylines, args.lights, seed=seeds[0])
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    return wm, full


def scene_on_device(tb, wm, args, dev, seed):
    """A synthetic scene batch of this workload's shape (seed = scene id), pre-processed, resident in HBM."""
    batch = tb.synthetic.make_scene(args.scenes, args.agents, args.polylines, args.lights, seed=seed)
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    return wm.pre_processing({k: v.to(dev) for k, v in full.items()})


_LATENT = {}


def _latent(n, A, dev):
    """The injected std-normal latent sample [n, A, 16] (seed 0), resident on the device: one host-to-device copy per shape, not one
    per scene (a blocking copy inside a loop over scenes would wait for the previous scene's rollout)."""
    key = (n, A, str(dev))
    if key not in _LATENT:
        _LATENT[key] = torch.randn(n, A, 16, generator=torch.Generator().manual_seed(0)).to(dev)
    return _LATENT[key]


def engine_inputs(wm, bd, args, dev, n_step):
    """Once-per-scene work (map encoder, traffic-light pre-compute, K/V tables) + the arguments of RolloutEngine.reset / refill."""
    R = args.rollouts
    mp, tl = wm.encode_scene(bd, n_rollout=R)
    r = (lambda t: t.repeat_interleave(R, 0)) if R > 1 else (lambda t: t)
    n, A = args.scenes * R, args.agents
    z = _latent(n, A, dev)  # prior sample (std-normal), injected
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
    eng._bench_scene = (bd, kw)  # (rule_check_leg: the scene batch the rule checker's map / light tables come from)
    return eng, t_scene


def rule_check_leg(a, wm, eng, use_graph, barrier, reduce_max):
    """The timed region once more WITH what the reference runs on every step of an inference rollout (waymo_motion.py:250:
    TrafficRuleChecker.check) and what consumes it (`_filter_futures`, data_modules/wosac_post_processing.py:31-64, when more than 32
    futures were rolled): W prime steps untimed, then K closed-loop steps + RolloutEngine.buffer(rule_checker=...) - the five
    metric-only checks over the whole device-resident log as ONE tbx_rule_check launch (+ tbx_rule_accumulate), the log handed out as
    the reference's RolloutBuffer - + the filter, all inside the clock. -> dict(value, ms_per_step, rule_ms, ...)."""
    from importlib import import_module as im

    bd, kw = eng._bench_scene
    R = a.rollouts
    PP = im("trafficbots_amd.data_modules.wosac_post_processing")
    post = PP.WOSACPostProcessing(step_gt=90, step_current=a.warmup, const_vel_z_sim=True, const_vel_no_sim=True, w_road_edge=0.5, use_wosac_col=True)
    role = bd.get("ref/ag_role")
    if role is None:
        role = torch.ones(a.scenes, a.agents, 3, dtype=torch.bool, device=eng.dev)
    dts, rule_ms, viol = [], [], None
    for rep in range(max(1, a.repeats)):
        eng.restore()
        checker = wm._rule_checker(bd, kw["ag_navi"], kw["tl_tokens"], n_rollout=R)
        checker._setup()  # (the compacted road-edge / lane tables: once per scene, like the map encoder - not per rollout)
        eng.run(a.warmup, use_graph=use_graph)
        barrier()
        t0 = time.perf_counter()
        eng.run(a.steps, use_graph=use_graph)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        buf = eng.buffer(a.warmup, rule_checker=checker)
        kept = None
        if R > post.n_joint_future:
            buf.flatten_joint_future(R)
            kept = post._filter_futures(buf, role)
        e1.record()
        barrier()
        dts.append(reduce_max(time.perf_counter() - t0))
        rule_ms.append(e0.elapsed_time(e1))
        viol = {k: float(v.float().mean()) for k, v in buf.violation.items() if not k.endswith("_this_step")}
        if kept is not None:
            viol["kept_futures"] = int(kept.shape[1])
    eng.restore()
    dt = sorted(dts)[len(dts) // 2]
    world = int(os.environ.get("WORLD_SIZE", 1))
    units = world * a.scenes * R * a.agents * a.steps
    return {"value": units / dt, "ms_per_step": dt / a.steps * 1e3, "rule_checks_ms": sorted(rule_ms)[len(rule_ms) // 2],
            "filter_futures": R > post.n_joint_future, "flag_rates": viol,
            "note": "K closed-loop steps + RolloutEngine.buffer(rule_checker): tbx_rule_check over every (rollout, step) frame of the log + "
                    "tbx_rule_accumulate (+ tbx_filter_futures for > 32 futures), timed together; rule_checks_ms = device time of that tail"}


def measure(a, tb, hip, dev, rank, world, dist):
    """-> (result dict, wm, full). The result holds `value`, `ms_per_step`, the repeats, `config`, `roofline`, `kernels`, the
    end-to-end figures; report.py decides what of it goes on the judged line."""
    E = import_module("trafficbots_amd.engine")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wm, full = build(tb, a, dev, rank)
    # (a timed region shorter than --graph-steps: one graph of all of it - the engine captures the multi-step graph for both
    # parities of the light tables' double buffer, so an odd number of warm-up steps does not split it)
    gsteps = max(1, min(a.graph_steps, a.steps) // 2 * 2)
    # this measurement's schedule belongs to its module / engine (engine.Schedule), not to the process
    wm.schedule = E.DEFAULT.replace(kv_bf16=bool(a.kv_bf16), lights_ahead=not a.no_lights_ahead, graph_steps=gsteps)
    if getattr(a, "attn_mfma", None) is not None:
        wm.schedule = wm.schedule.replace(attn_mfma=bool(a.attn_mfma))
    if a.kv_bf16 and getattr(a, "attn_mfma", None):  # = Schedule.reduced(): the bf16-arithmetic schedule (--linear-bf16 0: its LINEAR stages fp32-class)
        wm.schedule = wm.schedule.replace(linear_bf16=bool(getattr(a, "linear_bf16", 1)))
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
    if getattr(a, "rule_checks", False):
        def reduce_max(x):
            if world > 1:
                t = torch.tensor([x], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            return x

        try:
            with E.use(wm.schedule):
                wr = rule_check_leg(a, wm, eng, use_graph, barrier, reduce_max)
            wr["vs_unchecked"] = wr["value"] / timing["value"]
            timing["with_rule_checks"] = wr
        except Exception as e:  # noqa: BLE001 - the headline must still be printed
            timing["with_rule_checks"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ---- scene-to-scene reuse (the reference's validation_step loops over scenes, waymo_motion.py:526): NEW scenes through the
    # same engine - once-per-scene encoders + RolloutEngine.refill (in place: the captured graphs stay valid) + the W prime and K
    # closed-loop steps, everything timed; scene tensors resident in HBM as in the headline. No graph capture in this loop.
    reuse = None
    if use_graph and a.new_scenes > 0 and world == 1:
        first = shard_scenes(a.scenes * world, rank, world)[0]
        n_all = a.warmup + a.steps + 2 * a.profile_steps
        bds = [scene_on_device(tb, wm, a, dev, first + 1000 + i) for i in range(a.new_scenes)]
        with E.use(wm.schedule):
            # two untimed scenes first (refill + rollout): the first refills of a process pay one-time costs (allocator growth;
            # measured 90 ms, once, in the first OR the second refill) - the figure is the steady state of a loop over scenes
            for w_ in range(2):
                eng.refill(**engine_inputs(wm, scene_on_device(tb, wm, a, dev, first + 998 + w_), a, dev, n_all))
                eng.run(a.warmup + a.steps, use_graph=True)
        torch.cuda.synchronize()
        # (a) attribution: per new scene, the host + device time of [encoders + refill] alone - a device sync on both sides
        t_enc = []
        with E.use(wm.schedule):
            for bd in bds[:min(4, len(bds))]:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.refill(**engine_inputs(wm, bd, a, dev, n_all))
                torch.cuda.synchronize()
                t_enc.append(time.perf_counter() - t0)
                eng.run(a.warmup + a.steps, use_graph=True)
            torch.cuda.synchronize()
        # (b) end to end: the loop a caller runs - nothing in it waits for the device (the K-nearest / light-sharing checks stay on
        # the device), so the host prepares scene k + 1 (eager encoders + refill, enqueued behind) while the device rolls scene k
        # (b) end to end: the loop a serving caller runs (pl_modules/scene_loader.SceneLoader): per scene ~20 in-place copies of the
        # raw scene tensors into static inputs + the captured [encoders + derived state] on a side stream beside the previous scene's
        # rollout, then the captured [copies + K/V tables + priming] and the step graphs on the launch stream. Nothing in it waits
        # for the device.
        SL = import_module("trafficbots_amd.pl_modules.scene_loader")
        t_cap_refill = time.perf_counter()
        with E.use(wm.schedule):
            loader = SL.SceneLoader(eng, bds[0], lambda sb: engine_inputs(wm, sb, a, dev, n_all))
        torch.cuda.synchronize()
        t_cap_refill = time.perf_counter() - t_cap_refill
        t_load = []
        with E.use(wm.schedule):
            for bd in bds[:min(4, len(bds))]:  # attribution: a load alone, device idle on both sides
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loader.load(bd)
                torch.cuda.synchronize()
                t_load.append(time.perf_counter() - t0)
                eng.run(a.warmup + a.steps, use_graph=True)
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            loader.prefetch(bds[0])
            for i, bd in enumerate(bds):
                loader.commit()
                if i + 1 < len(bds):
                    # side stream, beside this scene's rollout. Enqueued BEFORE the rollout's replays: a hipGraph replay call returns
                    # only when the graph launched before it is nearly done (tools/scene_loop_profile.py: the host is never more than
                    # one replay ahead), so behind run() the prefetch would start when the rollout is all but over
                    loader.prefetch(bds[i + 1])
                eng.run(a.warmup + a.steps, use_graph=True)
            torch.cuda.synchronize()
        t_all = time.perf_counter() - t_all
        reuse = {"scenes": a.new_scenes, "new_scene_ms": sorted(t_load)[len(t_load) // 2] * 1e3, "new_scene_ms_all": [t * 1e3 for t in t_load],
                 "new_scene_eager_ms": sorted(t_enc)[len(t_enc) // 2] * 1e3, "refill_graph_capture_ms": t_cap_refill * 1e3,
                 "end_to_end_value": a.new_scenes * a.scenes * a.rollouts * a.agents * a.steps / t_all,
                 "ms_per_scene": t_all / a.new_scenes * 1e3,
                 "note": "per new scene: map encoder + light pre-compute + K/V tables + RolloutEngine.refill as ONE captured graph "
                         "(SceneLoader.load; new_scene_ms: measured alone, device idle on both sides; new_scene_eager_ms: the same work as "
                         "eager launches), then W prime + K closed-loop steps on the graphs captured once for this shape; end_to_end_value "
                         "counts the K steps' agent-steps over ALL of a loop over new scenes"}
        eng.restore()
    workload = {"workload": f"{a.agents}-agent/{a.polylines}-polyline/{a.lights}-light synthetic scene, "
                            f"{a.warmup}-step teacher-forced prime + {a.steps}-step closed-loop rollout",
                "scenes_per_gpu": a.scenes, "rollouts_per_scene": a.rollouts, "graph": use_graph,
                "steps_per_graph_replay": gsteps if use_graph else 0,
                "pre_roll_rollouts": n_pre,  # untimed whole-rollout replays before the W warm-up steps (device at steady clocks)
                "lights_one_step_ahead_on_second_stream": (not a.no_lights_ahead) and not eng.one_queue,
                "one_queue_paired_launches": bool(eng.one_queue),  # Schedule.one_queue: 5 paired launches per step, the lights one step ahead in the same grids
                "attn_mfma": wm.schedule.attn_mfma,
                "weights": "random init of the 10,657,094-parameter default architecture"}
    if a.profile_steps <= 0:  # tooling only (timeline traces, A/B runs, the scenes-per-GPU curve): no per-kernel timing pass
        return {**timing, "config": workload, "roofline": None, "kernels": None,
                "note": "--profile-steps 0: no per-kernel timing pass, not a judged line",
                "scene_encode_ms": t_scene * 1e3, "graph_capture_ms": t_cap * 1e3,
                "finite": bool(torch.isfinite(eng.S["out_pose"]).all())}, wm, full
    # ---- live per-kernel timing: eager steps right after the timed region, same state and the SAME launches as the timed
    # schedule, events on the launch stream, in the engine's one-stream order so that a kernel's duration is its own (in the
    # timed region the light and agent halves share the device, which stretches the kernels of both)
    # (a one-queue engine already IS one stream: its paired launches are timed as they run in the timed region)
    if not eng.one_queue:
        eng.sched = eng.sched.replace(lights_ahead=False)
    # the host must be AHEAD of the device while the events are recorded: an event pair around a launch otherwise also
    # times the wait for the host to enqueue that launch (seen on a loaded box: 27 us "launches" of a 10 us kernel).
    # A device-side delay in front lets the host queue all launches of the profiled steps first.
    torch.cuda._sleep(int(2.4e9 * (0.01 + 0.006 * a.profile_steps)))
    with events.KernelEvents(hip) as ke:
        eng.run(a.profile_steps, use_graph=False)
    classes = ke.classes(a.profile_steps)
    products = int(getattr(eng.sched, "mfma_products", 3))
    kernels = [events.kernel_entry(a, c, products) for c in classes]
    # the judged object: the kernel class the largest share of the step's kernel time goes to
    roof = dict(next(k for k in kernels if k["bound"] != "latency"))
    cnt = events.attn_counters(a)
    att = next((k for k in kernels if k["class"] == "attn" and k["source_rows_per_launch"] >= 1024), None)
    if cnt and att is not None:
        events.attach_attn_counters(att, cnt)
        if roof.get("class") == "attn" and roof.get("source_rows_per_launch") == att.get("source_rows_per_launch"):
            roof = dict(att)  # (the judged object was copied before the counters were attached)
    res = {
        **timing,
        "config": workload,
        "roofline": roof,
        "kernels": kernels,  # every kernel class of the step, largest share first (roofline = the first non-elementwise one)
        "roofline_gemm": events.gemm_summary(kernels),
        "scene_encode_ms": t_scene * 1e3, "graph_capture_ms": t_cap * 1e3,
        # SURVEY §8d "end-to-end": the once-per-scene work (map encoder, traffic-light pre-compute, K/V tables, engine refill)
        # counted into the same units, over new scenes rolled through the SAME engine; the first scene of a process additionally
        # pays scene_encode_ms (cold: allocations, weight packing) and graph_capture_ms once per shape
        "end_to_end_value": reuse["end_to_end_value"] if reuse else units / (dt + t_scene),
        "scene_reuse": reuse,
        "first_scene_value": units / (dt + t_scene + t_cap),
        "finite": bool(torch.isfinite(eng.S["out_pose"]).all()),
    }
    return res, wm, full
