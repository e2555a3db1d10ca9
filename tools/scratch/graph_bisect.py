"""Bisect the replay-vs-eager gradient mismatch of GraphedTrainStep on a small config."""
import sys, time, faulthandler, resource
faulthandler.enable()
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
tb = load_package()
hip = import_module("trafficbots_amd.hip"); hip.load()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "thread"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
big = len(sys.argv) > 3 and sys.argv[3] == "big"
if "nomt" in sys.argv:
    torch.autograd.set_multithreading_enabled(False)
cfg = tb.config.default_model_cfg() if big else tb.config.default_model_cfg(n_tgt_knn=4)
scfg = tb.config.default_sim_cfg()
cfg["tf_cfg"]["dropout_p"] = 0.0
cfg["mp_encoder"]["pl_encoder"]["mlp_dropout_p"] = 0.0
cfg["add_navi_latent"]["mlp_dropout_p"] = 0.0
scfg["teacher_forcing_training"]["prob_forcing_agent"] = 0.0
scfg["pre_processing"]["scene_centric"]["dropout_p_history"] = -1.0
scfg["time_step_end"] = steps
torch.manual_seed(0)
wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **scfg).to(dev).train()
(opt,), _ = wm.configure_optimizers()
shape = (2, 64, 1024, 128) if big else (2, 8, 64, 8)
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(*shape, seed=0).items()}
names = {id(p): k for k, p in wm.model.named_parameters()}
if mode == "main":
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    # capture on the main thread: monkey-patch the thread runner
    import threading
    class _T:
        def __init__(self, target, name=None): self.t = target
        def start(self): self.t()
        def join(self): pass
    DP.threading.Thread = _T
if "pre" in sys.argv:   # eager optimizer steps before the capture
    for _ in range(2):
        DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
    wm.last_metrics = None; wm.logged.clear(); opt.zero_grad(set_to_none=True)
gs = DP.GraphedTrainStep(wm, opt, batch, warmup=1, verbose=True)
if "post" in sys.argv:  # graphed optimizer steps (replay + clip + AdamW) before the comparison
    for _ in range(3):
        gs(batch)
if "postnoopt" in sys.argv:
    gs.opt = torch.optim.SGD(gs.live, lr=0.0)
    for _ in range(3):
        gs(batch)
def grads(): return [g.clone() for g in gs.grads]
gs.graph.replay(); torch.cuda.synchronize(); g1 = grads()
gs.graph.replay(); torch.cuda.synchronize(); g2 = grads()
def eager():
    for p in gs.live: p.grad = None
    loss = wm.training_step({k: v.clone() for k, v in batch.items()}, 0, noise=gs.noise, use_prior=gs.use_prior)
    loss.backward()
    return float(loss.detach()), [p.grad.clone() for p in gs.live]
le, ge = eager()
print("loss graph", float(gs.metrics["loss"].detach()), "eager", le)
def nbad(A, B):
    bad = [(names[id(p)], float((a - b).abs().max())) for a, b, p in zip(A, B, gs.live) if float((a - b).abs().max()) > 1e-3 * max(1.0, float(b.abs().max()))]
    return len(bad), bad[:6]
print("replay1 vs replay2:", nbad(g1, g2))
print("replay1 vs eager  :", nbad(g1, ge))
print("n live", len(gs.live))
