import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
hb = import_module("trafficbots_amd.hip_base")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
wm.train_precision = "bf16"
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
step = lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
step(); step()
lib = hb.load()
cnt = {"single": 0, "multi": 0, "jobs": 0}
s0, m0 = lib.tbx_pack_weight_mfma32, lib.tbx_pack_weight_mfma32_multi
def single(*a):
    cnt["single"] += 1
    return s0(*a)
def multi(arr, n, st):
    cnt["multi"] += 1; cnt["jobs"] += n
    return m0(arr, n, st)
lib.tbx_pack_weight_mfma32, lib.tbx_pack_weight_mfma32_multi = single, multi
step()
torch.cuda.synchronize()
print(cnt, "plan sizes", {k: len(v) for k, v in wm.model.__dict__.get("_tbx_pack_plans", {}).items()})
