"""Two independent RolloutEngines of R/2 rollouts each on their own streams vs one engine of R rollouts (WOSAC shape): do the launches
of one fill the other's gaps?  python tools/scratch/two_engines.py [R=32] [G=2]"""
import sys, time
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench as B
from __graft_entry__ import load_package
tb = load_package()
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
E = import_module("trafficbots_amd.engine")
Eng = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine


def make(rollouts, seed):
    sys.argv = ["x", "--no-cpu-baseline", "--agents", "128", "--rollouts", str(rollouts), "--steps", "40"]
    a = B.parse()
    wm, full = B.build(tb, a, dev, 0)
    wm.schedule = E.DEFAULT.replace(graph_steps=40)
    eng, _ = B.gpu_rollout_setup(tb, wm, full, a, dev)
    return eng


def timed(engs, streams, reps=5):
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.capture()
    torch.cuda.synchronize()
    best = []
    for rep in range(reps + 2):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.restore()
                e.run(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.run(40)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep >= 2:
            best.append(dt)
    return sorted(best)[len(best) // 2]


one = make(R, 0)
dt1 = timed([one], [torch.cuda.Stream()])
print(f"1 engine  x {R} rollouts: {R * 128 * 40 / dt1 / 1e6:.3f} M agent-steps/s ({dt1 / 40 * 1e3:.4f} ms per step)")
del one
torch.cuda.empty_cache()
engs = [make(R // G, i) for i in range(G)]
dtg = timed(engs, [torch.cuda.Stream() for _ in range(G)])
print(f"{G} engines x {R // G} rollouts: {R * 128 * 40 / dtg / 1e6:.3f} M agent-steps/s ({dtg / 40 * 1e3:.4f} ms per step)")
