"""Experiment: eager vs graph-captured training step (time, parity of the replayed loss / gradients)."""
import sys, time, faulthandler
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
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
drop0 = len(sys.argv) > 2 and sys.argv[2] == "parity"
cfg = tb.config.default_model_cfg()
scfg = tb.config.default_sim_cfg()
if drop0:
    cfg["tf_cfg"]["dropout_p"] = 0.0
    cfg["mp_encoder"]["pl_encoder"]["mlp_dropout_p"] = 0.0
    cfg["add_navi_latent"]["mlp_dropout_p"] = 0.0
    scfg["teacher_forcing_training"]["prob_forcing_agent"] = 0.0
    scfg["pre_processing"]["scene_centric"]["dropout_p_history"] = -1.0
torch.manual_seed(0)
wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **scfg).to(dev).train()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(scenes, 64, 1024, 128, seed=0).items()}
def sync(): torch.cuda.synchronize()
live = None
for i in range(2):
    sync(); t0 = time.perf_counter()
    DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()}, live=live)
    live = live or DP.live_parameters(wm.model)
    sync(); print("eager step", i, time.perf_counter() - t0, flush=True)
import warnings
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as wl:
    warnings.simplefilter("always")
    opt.zero_grad(set_to_none=True)
    nz, up = torch.zeros(scenes, 64, 16, device=dev), torch.zeros((), dtype=torch.bool, device=dev)
    bpre = {k: v for k, v in wm.pre_processing({k: v.clone() for k, v in batch.items()}).items() if torch.is_tensor(v)}
    n0 = len(wl)
    wm.training_step(dict(bpre), 0, noise=nz, use_prior=up).backward()
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter((str(w.filename).split("/")[-1], w.lineno) for w in wl[n0:])
print("sync sites in fwd+bwd:", c.most_common(20), flush=True)
t0 = time.perf_counter()
wm.last_metrics = None; wm.logged.clear(); opt.zero_grad(set_to_none=True)
gs = DP.GraphedTrainStep(wm, opt, batch, verbose=True)
sync(); print("capture (incl 2 warm-ups)", time.perf_counter() - t0, flush=True)
for i in range(4):
    sync(); t0 = time.perf_counter()
    m = gs(batch)
    sync(); print("graph step", i, time.perf_counter() - t0, float(m["loss"]), flush=True)
if drop0:
    # same noise / choice, eager vs replay
    gs.opt = torch.optim.SGD(gs.live, lr=0.0)
    torch.manual_seed(5); m = gs(batch); lg = float(m["loss"]); gg = [p.grad.clone() for p in gs.live]
    opt.zero_grad(set_to_none=True)
    loss = wm.training_step({k: v.clone() for k, v in batch.items()}, 0, noise=gs.noise, use_prior=gs.use_prior)
    loss.backward()
    print("loss graph", lg, "eager", float(loss))
    print("max grad diff", max(float((a - p.grad).abs().max()) for a, p in zip(gg, gs.live)), "max grad", max(float(a.abs().max()) for a in gg))
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
