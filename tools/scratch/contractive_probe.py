"""Per-step max |dpose| of the HIP closed loop vs the oracle for residual-branch scales (which scale makes the loop contractive)."""
import sys
from importlib import import_module
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package

tb = load_package()
from oracle import trafficbots_oracle as O
from test_hip_rollout import _oracle_tokens, _setup

dev = torch.device("cuda:0")
for scale, keys in ((0.3, ("linear2", "out_proj")), (0.1, ("linear2", "out_proj")), (0.3, ("linear2", "out_proj", "action")), (1.0, ())):
    wm, P, b, bd = _setup(tb, dev, (8, 64, 8), 4)
    with torch.no_grad():
        for k, p in wm.model.state_dict().items():
            hit = (("linear2" in keys and k.endswith(("linear2.weight", "linear2.bias"))) or ("out_proj" in keys and k.endswith(("out_proj_weight", "out_proj_bias")))
                   or ("action" in keys and k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k))
            if hit:
                p.mul_(scale); P[k] = P[k] * scale
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=4), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    mp_o, tl_o = _oracle_tokens(om, b)
    z = torch.randn(1, 8, 16, generator=torch.Generator().manual_seed(0))
    valid = b["sc/ag_valid"].any(-1)
    bh = dict(b); bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    with torch.no_grad():
        ro = O.Sim(om, scfg, False).rollout(bh, mp_o, tl_o, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, 90, gt_prefix="hist", tl_gt_key="sc/tl_state")
    mp, tl = wm.encode_scene(bd)
    ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["sc/ag_valid"], "gt_pose": bd["sc/ag_pose"],
                 "gt_motion": bd["sc/ag_motion"], "ag_latent": z.to(dev), "ag_latent_valid": valid.to(dev), "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid.to(dev)}
    buf = wm.rollout(ag_tokens, mp, tl, bd["sc/tl_state"], wm.teacher_forcing_joint_future_pred, wm._rule_checker(bd, bd["gt/ag_navi"], tl), 90, True)
    buf.flatten_joint_future(1)
    d = (buf.pred_pose[:, 0].cpu() - ro["pred_pose"]).abs().amax((0, 1, 3))
    veq = (buf.pred_valid[:, 0].cpu() == ro["pred_valid"]).all(0).all(0)
    print(scale, keys, "action max", float(ro["action"].abs().max()), "dpose@", [f"{float(d[t]):.1e}" for t in (10, 20, 30, 45, 60, 75, 89)], "valid_eq_until", int((~veq).float().argmax()) if (~veq).any() else 90, flush=True)
