#!/usr/bin/env python
"""Timing of the traffic-rule kernels over a whole rollout log: tbx_rule_check + tbx_rule_accumulate
(+ tbx_filter_futures) on a crowded synthetic episode of the WOSAC shape (32 rollouts x 128 agents x 80 steps, 1024
polylines, 128 lights) and of the configs[1] scene (1 x 64 x 80).

    python tools/rule_bench.py [--reps 20]
"""
import argparse
import json
import sys
from importlib import import_module
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    hip.load()
    T = import_module("trafficbots_amd.utils.traffic_rule_checker")
    dev = torch.device("cuda:0")
    out = []
    for name, n_roll, n_ag in (("configs[1] scene", 1, 64), ("WOSAC shape", 32, 128)):
        e = tb.synthetic.make_rule_episode(n_sc=1, n_ag=n_ag, n_mp=1024, n_tl=128, n_step=80, seed=5, extent=120.0)
        d = lambda t: t.to(dev)
        r = lambda t: t.repeat_interleave(n_roll, 0).to(dev)
        rc = T.TrafficRuleChecker(mp_boundary=d(e["map/boundary"]), mp_valid=d(e["map/valid"]), mp_type=d(e["map/type"]), mp_pos=d(e["map/pos"]),
                                  mp_dir=d(e["map/dir"]), ag_type=r(e["agent/type"]), ag_size=r(e["agent/size"]), ag_goal=None, ag_dest=None,
                                  tl_valid=r(e["tl/valid"]), tl_pose=r(e["tl/pose"]), disable_check=False)
        w = 1 << torch.arange(5, dtype=torch.int32)
        bits = (e["tl/state"].to(torch.int32) * w).sum(-1).to(torch.uint8)
        valid, pose, motion, tl = r(e["agent/valid"]).to(torch.uint8).contiguous(), r(e["agent/pose"]).contiguous(), r(e["agent/motion"]).contiguous(), r(bits).contiguous()
        ctx = rc._setup()
        n, A, Tn = valid.shape
        flags = torch.zeros(n, A, Tn, dtype=torch.uint8, device=dev)
        acc = torch.zeros_like(flags)
        run = lambda: hip.rule_check(ctx, valid, pose, motion, tl, Tn, 0, Tn, flags)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        t_check = e0.elapsed_time(e1) / a.reps * 1e3
        st, cnt = torch.zeros(n, A, dtype=torch.uint8, device=dev), torch.zeros(n, A, device=dev)
        e0.record()
        for _ in range(a.reps):
            hip.rule_accumulate(flags, n * A, Tn, 0, Tn, st, cnt, flags.clone(), acc)
        e1.record()
        torch.cuda.synchronize()
        n_seg, n_lane = int(ctx_count(rc, "n_seg")), int(ctx_count(rc, "n_lane"))
        frames = n * Tn
        out.append({"workload": f"{name}: {n} rollouts x {A} agents x {Tn} steps, 1024 polylines ({n_seg} road-edge segments, {n_lane} lane-centre nodes), 128 lights",
                    "rule_check_us": t_check, "rule_accumulate_us": e0.elapsed_time(e1) / a.reps * 1e3,
                    "agent_frames_per_s": frames * A / (t_check * 1e-6), "us_per_rollout_step": t_check / frames,
                    "flag_counts": {k: int(((flags & b) != 0).sum()) for k, b in (("collided", 1), ("collided_wosac", 2), ("run_road_edge", 4), ("run_red_light", 8), ("passive_raw", 16))}})
    print(json.dumps(out))


def ctx_count(rc, key):
    return rc._keep[key].sum().item()


if __name__ == "__main__":
    main()
