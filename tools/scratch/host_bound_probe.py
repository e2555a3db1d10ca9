"""Is the graph replay loop host-bound? Host time to enqueue N replays vs the device time of the same N replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from __graft_entry__ import load_package

sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tb = load_package()
wm, full = bench.build(tb, args, dev, 0)
eng, _ = bench.gpu_rollout_setup(tb, wm, full, args, dev)
eng.run(args.warmup, use_graph=True)
torch.cuda.synchronize()
N = 40
for rep in range(3):
    t0 = time.perf_counter()
    eng.run(N, use_graph=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue of {N} replays: {(t1 - t0) / N * 1e6:.1f} us/replay; until the device is done: {(t2 - t0) / N * 1e6:.1f} us/replay")
