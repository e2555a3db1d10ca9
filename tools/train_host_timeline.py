"""Where does the HOST wait inside GraphedTrainStep.__call__? perf_counter around every statement of 10 steps (a statement that takes
~the step's GPU time is a synchronisation point: the device then idles between steps while the host catches up), and the GPU time of the
replay alone (events) against the wall time per step.   python tools/train_host_timeline.py"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
wm.train_precision = "bf16"
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
gs = DP.GraphedTrainStep(wm, opt, batch)
for _ in range(3):
    gs(batch)
torch.cuda.synchronize()
acc = {}
ev = []


def T(name, fn, *a, **kw):
    t0 = time.perf_counter()
    r = fn(*a, **kw)
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


N = 10
seg = []
t_all = time.perf_counter()
for _ in range(N):
    s0, s1, s2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s0.record()
    b = T("_pre", gs._pre, batch)
    T("static copies", lambda: [v.copy_(b[k]) for k, v in gs.static.items()])
    T("_refill", gs._refill)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s1.record()
    e0.record()
    T("graph.replay", gs.graph.replay)
    e1.record()
    ev.append((e0, e1))
    T("flat.attach", gs.flat.attach)
    T("allreduce", DP.allreduce_gradients, gs.flat)
    T("clip", DP.clip_gradients, gs.flat, gs.clip)
    T("opt.step", gs.opt.step)
    s2.record()
    seg.append((s0, s1, e1, s2))
torch.cuda.synchronize()
print(f"device time before the replay (pre-processing, static copies, refill) {sum(a.elapsed_time(b) for a, b, _, _ in seg) / N:.3f} ms, "
      f"behind it (gather, clip, AdamW) {sum(c.elapsed_time(d) for _, _, c, d in seg) / N:.3f} ms per step")
wall = (time.perf_counter() - t_all) / N * 1e3
print(f"wall per step {wall:.2f} ms; replay on the device {sum(a.elapsed_time(b) for a, b in ev) / N:.2f} ms")
for k, v in acc.items():
    print(f"   host time in {k:16s} {v / N * 1e3:8.3f} ms per step")
