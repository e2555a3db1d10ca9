"""Probe: which ingredient makes hipStreamEndCapture of the training step segfault - an initialised RCCL process group, or an
eager training step (+ optimizer state) before the capture. usage: python pg_graph_probe.py <pg:0|1> <eager:0|1> [<collective before capture:0|1>]"""
import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29618", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
pg, eager, coll = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if pg:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
cfg = tb.config.default_model_cfg(n_tgt_knn=4)
scfg = tb.config.default_sim_cfg()
scfg["time_step_end"] = 20
wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **scfg).to(dev).train()
if coll:
    print('broadcast', DP.broadcast_parameters(wm.model), flush=True)
    torch.cuda.synchronize()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(2, 8, 64, 8, seed=0).items()}
if eager:
    DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
    print("eager done", flush=True)
gs = DP.GraphedTrainStep(wm, opt, batch, warmup=1, verbose=True)
m = gs(batch)
print("OK", pg, eager, float(m["loss"]), flush=True)
if pg:
    dist.destroy_process_group()
