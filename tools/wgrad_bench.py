"""tbx_linear_wgrad timings at the shapes of the time-batched training step. TBX_WGRAD_WGS=<n> sets the workgroup target."""
import sys; sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
load_package()
hip = import_module('trafficbots_amd.hip'); hip.load()
dev = torch.device('cuda:0')
for rows, n, k in ((2027520, 128, 128), (2027520, 64, 128), (2027520, 128, 16), (1013760, 64, 128), (1048576, 128, 384), (184320, 640, 128), (184320, 128, 640),
                   (184320, 512, 128), (184320, 128, 512), (92160, 640, 128), (92160, 128, 128)):
    dy, x = torch.randn(rows, n, device=dev), torch.randn(rows, k, device=dev)
    for _ in range(3): hip.linear_wgrad(dy, x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.linear_wgrad(dy, x, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{rows:8d} x {n:4d} x {k:4d}: {ms:7.3f} ms  {rows * (n + k) * 4 / ms / 1e9:6.2f} TB/s  {2 * rows * n * k / ms / 1e9:6.1f} TF/s  splits {hip.load().tbx_linear_wgrad_splits(rows, n, k)}")
