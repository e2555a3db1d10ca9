"""Shapes, counts and forward times of the F.linear calls of one (eager) training step at configs[2]
(16 scenes x 64 agents x 1024 polylines x 128 lights, 90 steps), largest first. Usage: python tools/train_gemm_shapes.py [scenes]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(n, 64, 1024, 128, seed=0).items()}
DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
torch.cuda.synchronize()
stats = {}
orig = F.linear
def timed(x, w, b=None):
    rows = x.numel() // x.shape[-1]
    if rows < 20000:
        return orig(x, w, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    y = orig(x, w, b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    key = (rows, x.shape[-1], w.shape[0], torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))
    c = stats.setdefault(key, [0, 0.0]); c[0] += 1; c[1] += dt
    return y
F.linear = timed
torch.cuda.synchronize(); t0 = time.perf_counter()
DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
torch.cuda.synchronize(); print("step (eager, instrumented) s:", time.perf_counter() - t0)
F.linear = orig
print("| rows | K | N | grad | calls | fwd ms/call | fwd TF/s |")
for (rows, k, nn, g), (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"| {rows} | {k} | {nn} | {int(g)} | {c} | {t / c * 1e3:.3f} | {2 * rows * k * nn / (t / c) / 1e12:.1f} |")
print("peak memory GB:", torch.cuda.max_memory_allocated() / 2**30)
