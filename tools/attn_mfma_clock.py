#!/usr/bin/env python
"""Microbenchmark of tbx_knarpe_attn_fwd_mfma at the WOSAC shape (32 rollouts x 128 agents): the self attention
(25 targets out of the step's own 128 token rows) and the cross attention (64 of 1024 map tokens shared by the 32 rollouts + 25 of
128 lights), K-nearest sets of neighbouring rollouts nearly equal (as in a real scene). Event-timed against tbx_knarpe_attn_fwd
(VALU) on the same inputs. (The phase clock of the kernel's first form - s_memtime stamps around its four phases - is in
profiles/r04_attn_mfma_phase_clock.txt; the stamps went with that form.)
    python tools/attn_mfma_clock.py"""
import ctypes as C
import os
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

tb = load_package()
hip = import_module("trafficbots_amd.hip")
lib = hip.load()
from oracle import hptr_ops as H  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
n, S = 32, 128
rows = n * S


def knn_like(T, K, shared):
    """[n, S, K] int32: a source token's K targets; neighbouring rollouts (batch entries) differ in ~10 % of the slots"""
    base = torch.stack([torch.randperm(T, generator=g)[:K] for _ in range(S)])  # [S, K]
    idx = base[None].repeat(n, 1, 1)
    if shared:
        flip = torch.rand(n, S, K, generator=g) < 0.1
        idx = torch.where(flip, torch.randint(0, T, (n, S, K), generator=g), idx)
    else:
        idx = torch.stack([torch.stack([torch.randperm(T, generator=g)[:K] for _ in range(S)]) for _ in range(n)])
    return idx.to(torch.int32)


def seg(T, K, div, dtype, shared=True):
    kv = torch.randn((n // div) * T, 256, generator=g).to(dev).to(dtype)
    inv = (torch.rand(n, S, K, generator=g) < 0.1).to(torch.uint8).to(dev)
    rel = torch.cat([(torch.rand(n, S, K, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
    return hip.Seg(kv, 0, 128, T, knn_like(T, K, shared).to(dev), inv, None, div, rel=rel)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


q = (torch.randn(rows, 896, generator=g) * 0.5).to(dev)
bias = torch.zeros(128, device=dev)
out = torch.empty(rows, 640, device=dev)
flag = torch.empty(rows, dtype=torch.uint8, device=dev)
for dtype in (torch.bfloat16, torch.float32):
    cases = {"self  (25 of the step's 128 rows)": [seg(S, 25, 1, dtype, shared=False)],
             "cross (64 of 1024 map + 25 of 128 lights)": [seg(1024, 64, 32, dtype), seg(128, 25, 32, dtype)]}
    for name, segs in cases.items():
        pairs = rows * sum(s.k for s in segs)
        t_valu = timed(lambda: hip.knarpe_attn(q, 0, 384, bias, n, S, segs, out, flag, fxy, fyw))
        line = f"{str(dtype)[6:]:9s} {name:44s} {pairs / 1e3:6.0f} k pairs   VALU {t_valu:6.1f} us"
        t = timed(lambda: hip.knarpe_attn_mfma(q, 0, 384, n, S, segs, out, flag, fxy, fyw))
        line += f"   mfma {t:6.1f} us"
        print(line, flush=True)
