"""Max |engine - oracle| of the full-gain free rollout (tests/test_hip_rollout.py::test_free_rollout_80_steps_full_gain_contractive_weights)
per horizon, exact-fp32 schedule: a 1-ulp change anywhere in the loop shows up here as a different divergence curve.
usage: [TBX_HIP_LIB=...] python tools/scratch/free_rollout_probe.py"""
import sys
from importlib import import_module
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package

tb = load_package()
T = import_module("test_hip_rollout")
O = T.O
dev = torch.device("cuda:0")
wm, P, b, bd = T._setup(tb, dev, (8, 64, 8), 4)
with torch.no_grad():
    for k, p in wm.model.state_dict().items():
        if k.endswith(("linear2.weight", "linear2.bias", "out_proj_weight", "out_proj_bias")) or (k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k):
            p.mul_(0.3); P[k] = P[k] * 0.3
cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=4), tb.config.default_sim_cfg()
om = O.TrafficBotsOracle(P, cfg, training=False)
mp_o, tl_o = T._oracle_tokens(om, b)
z = torch.randn(1, 8, 16, generator=torch.Generator().manual_seed(0))
valid = b["sc/ag_valid"].any(-1)
bh = dict(b)
bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
with torch.no_grad():
    ro = O.Sim(om, scfg, False).rollout(bh, mp_o, tl_o, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, 90, gt_prefix="hist", tl_gt_key="sc/tl_state")
E = import_module("trafficbots_amd.engine")
for rep in range(2):
    wm.schedule = E.DEFAULT.replace(tile_small=bool(rep), dec_tail_mfma=bool(rep))
    mp, tl = wm.encode_scene(bd)
    ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["sc/ag_valid"],
                 "gt_pose": bd["sc/ag_pose"], "gt_motion": bd["sc/ag_motion"], "ag_latent": z.to(dev), "ag_latent_valid": valid.to(dev),
                 "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid.to(dev)}
    buf = wm.rollout(ag_tokens, mp, tl, bd["sc/tl_state"], wm.teacher_forcing_joint_future_pred, wm._rule_checker(bd, bd["gt/ag_navi"], tl), 90, True)
    buf.flatten_joint_future(1)
    dp = (buf.pred_pose[:, 0].cpu() - ro["pred_pose"]).abs().amax(dim=(0, 1, 3))
    dm = (buf.pred_motion[:, 0].cpu() - ro["pred_motion"]).abs().amax(dim=(0, 1, 3))
    print("rep", rep, "pose", " ".join(f"{float(dp[s]):.1e}" for s in (11, 15, 20, 30, 40, 50, 60, 69, 79, 89)))
    print("rep", rep, "motn", " ".join(f"{float(dm[s]):.1e}" for s in (11, 15, 20, 30, 40, 50, 60, 69, 79, 89)))
    import hashlib
    print("hash", hashlib.sha1(buf.pred_pose.cpu().numpy().tobytes()).hexdigest()[:12])
