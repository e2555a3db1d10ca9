"""Ad-hoc: a 12-scene x 64-agent rollout (768 rows: neither live-row nor large) on the default schedule vs the exact-fp32 schedule."""
import sys
from importlib import import_module
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

tb = load_package()
E = import_module("trafficbots_amd.engine")
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
dev = torch.device("cuda:0")
for n_sc, A in ((12, 64), (3, 64), (20, 40)):
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=16), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(dev).eval()
    batch = tb.synthetic.make_scene(n_sc, A, 256, 32, seed=1)
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    bd = wm.pre_processing({k: v.to(dev) for k, v in full.items()})
    outs = {}
    for name, sched in (("exact", E.DEFAULT.replace(dec_tail_mfma=False, tile_small=False, navi_rider=False)), ("default", E.DEFAULT)):
        wm.schedule = sched
        mp, tl = wm.encode_scene(bd, tl_valid_key="gt/tl_valid")
        z = torch.randn(n_sc, A, 16, generator=torch.Generator().manual_seed(1)).to(dev)
        valid = bd["gt/ag_valid"].any(-1)
        outs[name] = wm.reactive_replay(bd, mp, tl, z, valid, bd["gt/ag_navi"], valid, wm.teacher_forcing_joint_future_pred, True, step_end=14)
    a, b = outs["exact"], outs["default"]
    d = float((a.pred_pose[..., :12, :] - b.pred_pose[..., :12, :]).abs().max())
    da = float((a.vis_dict["action"][..., :11, :] - b.vis_dict["action"][..., :11, :]).abs().max())
    print(f"{n_sc} x {A}: rows {n_sc * A}: max |dpose| over 12 steps {d:.2e}, max |daction| over the 11 forced steps {da:.2e}, finite {bool(torch.isfinite(b.pred_pose).all())}")
    assert d < 5e-3 and da < 5e-3
print("OK")
