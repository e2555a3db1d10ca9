"""Stress: many rollouts (engine + 2 step graphs each) in one process - does replay survive?"""
import os, sys, gc, faulthandler
faulthandler.enable()
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
tb = load_package()
mode = sys.argv[1] if len(sys.argv) > 1 else "drop"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
keep = []
def one(i, n_sc):
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(dev).eval()
    batch = tb.synthetic.make_scene(n_sc, 8, 64, 8, seed=i)
    b = wm.pre_processing({k: v.to(dev) for k, v in {**batch, **tb.synthetic.to_history_batch(batch)}.items()})
    mp, tl = wm.encode_scene(b, tl_valid_key="gt/tl_valid")
    valid = b["gt/ag_valid"].any(-1)
    buf = wm.reactive_replay(b, mp, tl, torch.zeros(n_sc, 8, 16, device=dev), valid, b["gt/ag_navi"], valid,
                             wm.teacher_forcing_reactive_replay, True, step_end=12)
    torch.cuda.synchronize()
    if mode == "keep":
        keep.append((wm, buf))
    return float(buf.pred_pose.sum())
for i in range(N):
    v = one(i, 1 + i % 2)
    if mode == "gc":
        gc.collect()
    print(i, "ok", round(v, 2), flush=True)
print("survived", N)
