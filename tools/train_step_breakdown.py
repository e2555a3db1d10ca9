"""Where a GraphedTrainStep call spends its wall time: pre-processing + input copies, graph replay, gradient exchange + clip + AdamW."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
gs = DP.GraphedTrainStep(wm, opt, batch)
for _ in range(2): gs(batch)
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full call            %.1f ms" % t(lambda: gs(batch)))
print("graph replay         %.1f ms" % t(lambda: gs.graph.replay()))
def pre():
    b = gs._pre(batch)
    for k, v in gs.static.items(): v.copy_(b[k])
    gs._refill()
print("pre + copies + refill %.1f ms" % t(pre))
def post():
    DP.allreduce_gradients(gs.live)
    torch.nn.utils.clip_grad_norm_(gs.live, gs.clip)
    gs.opt.step()
print("clip + AdamW          %.1f ms" % t(post))
print("peak memory GB        %.1f" % (torch.cuda.max_memory_allocated() / 2**30))
