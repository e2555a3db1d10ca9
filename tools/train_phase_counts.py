"""Kernel launches and GPU time of the phases of one eager training step (configs[2]): which part of the step the replayed
graph spends its ~3 us per tiny launch on. Usage: python tools/train_phase_counts.py"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
TG = import_module("trafficbots_amd.train_graph")
torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
torch.cuda.synchronize()
marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
count = {"n": 0}
def wrap(name, fn):
    def f(*a, **k):
        marks.append((name + ":begin", ev()))
        r = fn(*a, **k)
        marks.append((name + ":end", ev()))
        return r
    return f
for name in ("map_encoder", "tl_pre_compute", "latent_posterior", "navi_predictor", "training_rollout", "policy_step", "tl_encoder", "training_loss"):
    setattr(TG, name, wrap(name, getattr(TG, name)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    marks.append(("step:begin", ev()))
    loss = wm.training_step({k: v.clone() for k, v in batch.items()}, 0)
    marks.append(("fwd:end", ev()))
    loss.backward()
    marks.append(("bwd:end", ev()))
    torch.cuda.synchronize()
kern = sorted([(e.time_range.start, e.time_range.end - e.time_range.start) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start], key=lambda x: x[0])
print("kernels total:", len(kern), "GPU busy ms:", sum(d for _, d in kern) / 1e3)
# phases by host order: count kernels launched between marks via event elapsed is launch-bound in eager mode; instead report
# per-phase kernel counts from the profiler's CPU-side ranges
import collections
tot = collections.OrderedDict()
t0 = marks[0][1]
prev_name, prev_t = None, 0.0
for name, e in marks:
    t = t0.elapsed_time(e)
    print(f"{name:28s} at {t:9.2f} ms (eager GPU timeline)")
