"""`cpu_baseline`: the oracle (CPU port of the reference formulation, test infrastructure under oracle/) timed on rank 0's host cores
on a bounded sample of the headline workload. This is the only place bench.py touches oracle/ - as the thing timed BESIDE the
product, never inside it."""
import time

import torch


def cpu_baseline(tb, wm, full, args):
    """Oracle on a bounded sample: first scene, whole `cpu_steps`-step closed-loop rollouts until ~10 s of CPU work."""
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
    return {"value": args.agents * n_done / dt, "unit": "sim-agent-steps/s", "cores": best[1], "kind": "port",
            "sample": f"1 scene x {args.agents} agents x {n_done} closed-loop steps ({n_done // args.cpu_steps} rollouts of {args.cpu_steps}) in {dt:.1f}s "
                      f"(oracle, torch {torch.__version__} CPU fp32, best of 8/16/32/64/all = {best[1]} threads of {n_all}, map encoding excluded)"}
