#!/usr/bin/env python
"""Host-side timeline of the loop over new scenes (no device sync inside): per scene the host time of [encoders], [refill], [run =
graph replays enqueued] - which call waits for the device? Then the same loop's wall time per scene.
    python tools/scene_loop_profile.py [bench.py rollout args]"""
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402
from tools.benchlib import rollout as R  # noqa: E402
from tools.benchlib.args import parse  # noqa: E402

a = parse()
tb = load_package()
hip = import_module("trafficbots_amd.hip")
hip.load()
E = import_module("trafficbots_amd.engine")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wm, full = R.build(tb, a, dev, 0)
wm.schedule = E.DEFAULT.replace(graph_steps=40)
eng, _ = R.gpu_rollout_setup(tb, wm, full, a, dev)
eng.capture()
n_all = a.warmup + a.steps + 2 * a.profile_steps
bds = [R.scene_on_device(tb, wm, a, dev, 1000 + i) for i in range(12)]
with E.use(wm.schedule):
    for bd in bds[:3]:
        eng.refill(**R.engine_inputs(wm, bd, a, dev, n_all))
        eng.run(a.warmup + a.steps, use_graph=True)
    torch.cuda.synchronize()
    rows = []
    t_all = time.perf_counter()
    for bd in bds:
        t0 = time.perf_counter()
        kw = R.engine_inputs(wm, bd, a, dev, n_all)
        t1 = time.perf_counter()
        eng.refill(**kw)
        t2 = time.perf_counter()
        eng.run(a.warmup + a.steps, use_graph=True)
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2))
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
print(f"{len(bds)} scenes: {t_all / len(bds) * 1e3:.2f} ms per scene wall, host loop returned after {t_host / len(bds) * 1e3:.2f} ms per scene")
print("host ms per scene:   encoders   refill   run (enqueue)")
for r in rows:
    print("                   " + "  ".join(f"{x * 1e3:8.2f}" for x in r))

# ---- the captured refill: device time of its two graphs alone, and a kernel count of each (torch profiler would add its own cost)
SL = import_module("trafficbots_amd.pl_modules.scene_loader")
with E.use(wm.schedule):
    loader = SL.SceneLoader(eng, bds[0], lambda sb: R.engine_inputs(wm, sb, a, dev, n_all))
    torch.cuda.synchronize()

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    print(f"graph_prepare (input copies excluded) {timed(eng.graph_prepare.replay):.3f} ms   graph_commit {timed(eng.graph_commit.replay):.3f} ms   "
          f"prime graph {timed(eng.graph_prime.replay):.3f} ms   one 40-step graph {timed((list(eng.graph_multi.values())[0] if isinstance(eng.graph_multi, dict) else eng.graph_multi).replay, 5):.3f} ms")
