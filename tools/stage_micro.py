#!/usr/bin/env python
"""Warm per-stage in-kernel cost of synthetic row-chain programs (profiling build, see tools/stage_clock.py)."""
import ctypes as C
import os
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TBX_HIP_LIB", str(ROOT / "trafficbotsv1.5_amd" / "csrc" / "libtbx_hip_clk.so"))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

load_package()
hip = import_module("trafficbots_amd.hip")
lib = hip.load()
lib.tbx_debug_clock_dump.argtypes = [C.c_void_p, C.c_int]
from trafficbots_amd.hip import AUX, BUF0, BUF1, Chain  # noqa: E402

OPS = {1: "LOAD", 2: "LINEAR", 3: "LN", 4: "ADD", 5: "COPY", 6: "ROWMASK", 7: "GROUPMAX", 8: "POOLMAX", 9: "STORE", 10: "CLAMP", 11: "DROPOUT", 12: "ATTN", 13: "ATTNSEG"}
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(rows, 640, device=dev)
out = torch.empty(rows, 1024, device=dev)
W = {k: torch.randn(*s, device=dev) * 0.05 for k, s in dict(w128=(128, 128), w512=(512, 128), w512b=(128, 512), w384=(384, 128)).items()}
b = {k: torch.randn(v.shape[0], device=dev) for k, v in W.items()}
g, be = torch.ones(128, device=dev), torch.zeros(128, device=dev)
mask = torch.zeros(rows, dtype=torch.uint8, device=dev)

ch = Chain(16, 1028)
ch.load(x[:, :128], BUF0, 0, n=128)
for _ in range(4):
    ch.add(BUF0, 0, BUF1, 0, 4)
for _ in range(4):
    ch.add(BUF0, 0, BUF1, 0, 128)
for _ in range(3):
    ch.layernorm(BUF0, 0, BUF1, 0, g, be)
for _ in range(3):
    ch.rowmask(BUF1, 0, 128, mask)
for _ in range(3):
    ch.linear(BUF0, 0, BUF1, 0, W["w128"], b["w128"])
ch.linear(BUF0, 0, BUF1, 0, W["w512"], b["w512"], relu=True)
ch.linear(BUF1, 0, BUF0, 0, W["w512b"], b["w512b"], accum=True)
ch.linear(BUF0, 0, BUF1, 0, W["w512"], b["w512"], relu=True)
ch.linear(BUF1, 0, BUF0, 0, W["w512b"], b["w512b"], accum=True)
ch.linear(BUF0, 0, BUF1, 0, W["w384"], b["w384"])
ch.store(BUF1, 0, 384, out)
ch.store(BUF1, 0, 128, out)
for _ in range(5):
    ch.run(rows)
torch.cuda.synchronize()
lib.tbx_debug_clock_reset()
ch.run(rows)
slots = hip.MAX_STAGES + 4
buf = (C.c_uint64 * (4 * slots))()
n = lib.tbx_debug_clock_dump(buf, 4)
for j, s in enumerate(ch.stages):
    print(f"{OPS[s.op]:9s} k={s.k:4d} n={s.n:4d}   {(buf[j + 1] - buf[j]) / 100.0:6.2f} us")
