"""Micro-benchmark: tbx_tall_linear vs F.linear at the time-batched pass's shapes."""
import sys, time
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
load_package()
hip = import_module("trafficbots_amd.hip")
dev = torch.device("cuda:0")
for m, k, n in ((184320, 128, 640), (184320, 640, 128), (92160, 128, 128), (184320, 128, 512), (184320, 512, 128), (184320, 128, 256), (2027520, 128, 128), (92160, 128, 640)):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    for name, fn in (("lib", lambda: torch.nn.functional.linear(x, w, b)), ("tall", lambda: hip.tall_linear(x, w, b))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"{m:8d} x {k:3d} -> {n:3d}  {name:5s} {dt * 1e6:8.1f} us  {2.0 * m * k * n / dt / 1e12:6.1f} TF/s  {(m * (k + n) * 4) / dt / 1e12:5.2f} TB/s")
