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
if "dist" in sys.argv:
    import os, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(4, device=dev); dist.all_reduce(t); print("nccl up", t.tolist(), flush=True)
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
wm.last_metrics = None; wm.logged.clear(); opt.zero_grad(set_to_none=True)
t0 = time.perf_counter()
gs = DP.GraphedTrainStep(wm, opt, batch, verbose=True, warmup=1)
sync(); print("capture (incl warm-up)", time.perf_counter() - t0, flush=True)
for i in range(3):
    sync(); t0 = time.perf_counter()
    m = gs(batch)
    sync(); print("graph step", i, time.perf_counter() - t0, float(m["loss"].detach()), flush=True)
if drop0:
    gs.opt = torch.optim.SGD(gs.live, lr=0.0); gs.clip = 0
    names = {id(p): k for k, p in wm.model.named_parameters()}
    def eager():
        for p in gs.live: p.grad = None
        loss = wm.training_step({k: v.clone() for k, v in batch.items()}, 0, noise=gs.noise, use_prior=gs.use_prior)
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in gs.live]
    le1, ge1 = eager(); le2, ge2 = eager()
    print("loss eager", le1, le2)
    def report(tag, A, B):
        worst = sorted(((float((a - b).abs().max()), float(a.abs().max()), names[id(p)]) for a, b, p in zip(A, B, gs.live)), reverse=True)[:4]
        print(tag, worst, flush=True)
    report("eager vs eager", ge1, ge2)
    gs.graph.replay(); torch.cuda.synchronize()   # same noise / choice as the eager runs: replay without refill
    gr = [g.clone() for g in gs.grads]
    print("loss graph (same noise)", float(gs.metrics["loss"].detach()))
    report("graph vs eager", gr, ge1)
    gs.graph.replay(); torch.cuda.synchronize()
    gr2 = [g.clone() for g in gs.grads]
    report("graph vs graph", gr, gr2)
    bad = []
    for a, b, p in zip(gr, ge1, gs.live):
        d = (a - b).abs()
        if float(d.max()) > 1e-3 * max(1.0, float(b.abs().max())):
            i = int(d.flatten().argmax())
            bad.append((names[id(p)], tuple(p.shape), int((d > 1e-3).sum()), float(a.flatten()[i]), float(b.flatten()[i]), float(gr2[len(bad)].flatten()[0]) if False else 0))
    print("n bad params", len(bad), "of", len(gs.live))
    for x in bad[:40]: print("   ", x)
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
