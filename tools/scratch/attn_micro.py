"""Attention kernel launch time at a few row counts (self 25 + cross 64 / 25 pairs), for A/B of kernel variants through env vars."""
import sys, time
from importlib import import_module
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
tb = load_package()
hip = import_module("trafficbots_amd.hip"); hip.load()
from oracle import hptr_ops as H
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
for n, S in ((16, 64), (32, 128), (8, 128), (64, 128)):
    rows = n * S
    q = torch.randn(rows, 640, generator=g).to(dev); bias = torch.randn(128, generator=g).to(dev)
    def seg(T, K, div):
        kv = torch.randn((n // div) * T, 256, generator=g).to(dev)
        idx = torch.randint(0, T, (n, S, K), generator=g).to(torch.int32).to(dev)
        inv = (torch.rand(n, S, K, generator=g) < 0.2).to(torch.uint8).to(dev)
        rel = torch.cat([(torch.rand(n, S, K, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        return hip.Seg(kv, 0, 128, T, idx, inv, None, div, rel=rel)
    cases = {"self25": [seg(S, 25, 1)], "cross89": [seg(1024, 64, 1), seg(128, 25, 1)]}
    out = torch.empty(rows, 640, device=dev); flag = torch.empty(rows, dtype=torch.uint8, device=dev)
    seed = torch.tensor([5], dtype=torch.int64, device=dev)
    for name, segs in cases.items():
        for drop in (None, (0.1, seed, 3)):
            for _ in range(5): hip.knarpe_attn(q, 0, 128, bias, n, S, segs, out, flag, fxy, fyw, drop=drop)
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): hip.knarpe_attn(q, 0, 128, bias, n, S, segs, out, flag, fxy, fyw, drop=drop)
            e1.record(); torch.cuda.synchronize()
            print(f"rows {rows:5d} {name:8s} drop={drop is not None!s:5s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
