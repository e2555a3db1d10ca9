import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from __graft_entry__ import load_package
tb = load_package()
from test_hip_rollout import _setup, O
dev = torch.device('cuda:0')
wm, P, b, bd = _setup(tb, dev, (8, 64, 8), 4)
cfg = tb.config.default_model_cfg(n_tgt_knn=4); scfg = tb.config.default_sim_cfg()
om = O.TrafficBotsOracle(P, cfg, training=False)
g = torch.Generator().manual_seed(0)
z = torch.randn(1, 8, 16, generator=g); valid = b["gt/ag_valid"].any(-1)
with torch.no_grad():
    mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
    tl_o = om.tl_pre_compute(b["gt/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    ro = O.Sim(om, scfg, False).rollout(b, mp_o, tl_o, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, 90)
mp, tl = wm.encode_scene(bd, tl_valid_key="gt/tl_valid")
buf = wm.reactive_replay(bd, mp, tl, z.to(dev), valid.to(dev), bd["gt/ag_navi"], valid.to(dev), wm.teacher_forcing_joint_future_pred, True, step_end=90, use_graph=False)
d = (buf.pred_pose[:, 0].cpu() - ro["pred_pose"]).abs().amax((0, 1, 3))
da = (buf.vis_dict["action"][:, 0].cpu() - ro["action"]).abs().amax((0, 1, 3))
print("pose err per step:", ["%.1e" % v for v in d.tolist()])
print("action err per step:", ["%.1e" % v for v in da.tolist()])
print("action magnitude:", ro["action"].abs().amax((0,1,3))[:20])
