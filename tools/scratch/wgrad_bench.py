"""Micro-benchmark + check of tbx_linear_wgrad at the time-batched pass's shapes (TBX_WGRAD_BF16=1: the split-bf16 form)."""
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
for m, k, n in ((184320, 128, 640), (184320, 640, 128), (92160, 128, 128), (184320, 128, 512), (2027520, 128, 128), (2027520, 128, 64), (1013760, 64, 64)):
    x = torch.randn(m, k, device=dev); dy = torch.randn(m, n, device=dev)
    fn = lambda: hip.linear_wgrad(dy, x, True)
    for _ in range(3): dw, db = fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    ref = dy[:20000].double().t() @ x[:20000].double()
    dw2, _ = hip.linear_wgrad(dy[:20000].contiguous(), x[:20000].contiguous(), True)
    bound = dy[:20000].abs().double().t() @ x[:20000].abs().double()
    err = float(((dw2.double() - ref).abs() / bound).max())
    print(f"{m:8d} x ({n:3d} <- {k:3d})  {dt * 1e6:8.1f} us  {2.0 * m * k * n / dt / 1e12:6.1f} TF/s  {(m * (k + n) * 4) / dt / 1e12:5.2f} TB/s   err/bound {err:.2e}  db err {float((db - dy.sum(0)).abs().max() / dy.abs().sum(0).max()):.1e}")
