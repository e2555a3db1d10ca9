"""Host profile of one new scene through a captured engine (bench.py's scene_reuse loop): engine_inputs + RolloutEngine.refill."""
import cProfile, pstats, sys, time, argparse
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench as B
from __graft_entry__ import load_package
tb = load_package()
dev = torch.device("cuda:0")
a = B.parse_args([]) if hasattr(B, "parse_args") else None
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).eval()
class A: pass
a = A(); a.rollouts, a.scenes, a.agents, a.polylines, a.lights, a.warmup, a.steps, a.profile_steps = 1, 1, 64, 1024, 128, 10, 80, 0
def scene(seed):
    batch = tb.synthetic.make_scene(1, 64, 1024, 128, seed=seed)
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    return wm.pre_processing({k: v.to(dev) for k, v in full.items()})
Eng = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine
eng = Eng(wm.model, wm.dynamics, dev, schedule=wm.schedule)
eng.reset(**B.engine_inputs(wm, scene(0), a, dev, 90))
eng.run(90)
bds = [scene(10 + i) for i in range(12)]
for bd in bds[:2]:
    eng.refill(**B.engine_inputs(wm, bd, a, dev, 90)); eng.run(90)
torch.cuda.synchronize()
ts = {"inputs": 0.0, "refill": 0.0}
for bd in bds[2:7]:
    t0 = time.perf_counter(); kw = B.engine_inputs(wm, bd, a, dev, 90); torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.refill(**kw); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts["inputs"] += (t1 - t0) / 5; ts["refill"] += (t2 - t1) / 5
    eng.run(90)
print({k: round(v * 1e3, 3) for k, v in ts.items()}, "ms per scene")
pr = cProfile.Profile()
for bd in bds[7:12]:
    kw = B.engine_inputs(wm, bd, a, dev, 90)
    pr.enable(); eng.refill(**kw); pr.disable()
    eng.run(90)
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
