#!/usr/bin/env python
"""Phase timing inside the attention sweep (profiling build libtbx_hip_clk.so: `make -C trafficbotsv1.5_amd/csrc clk`): wave 0 of
workgroup 0 sums the 100 MHz s_memtime ticks of every pass's phases.
    python tools/attn_clock.py"""
import ctypes as C
import os
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["TBX_HIP_LIB"] = str(ROOT / "trafficbotsv1.5_amd" / "csrc" / "libtbx_hip_clk.so")
RING = len(sys.argv) > 1 and sys.argv[1] == "ring"
os.environ["TBX_ATTN_RING_LONE_ROWS"] = "0"  # the plain sweep at every size ...
if RING:
    os.environ["TBX_ATTN_RING"] = "1"  # ... or the LDS-ring kernel (python tools/attn_clock.py ring)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

tb = load_package()
hip = import_module("trafficbots_amd.hip")
lib = hip.load()
lib.tbx_debug_attn_clock.argtypes = [C.c_void_p]
from oracle import hptr_ops as H  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
PH = ["loads of index / mask / pose -> embedding (8 sincos)", "-> scores (needs the K rows)", "-> softmax probabilities", "-> accumulate (needs the V rows)"]
for n, S in ((16, 64), (32, 128)):
    rows = n * S
    q = torch.randn(rows, 640, generator=g).to(dev)
    bias = torch.randn(128, generator=g).to(dev)

    def seg(T, K):
        kv = torch.randn(n * T, 256, generator=g).to(dev)
        idx = torch.randint(0, T, (n, S, K), generator=g).to(torch.int32).to(dev)
        inv = (torch.rand(n, S, K, generator=g) < 0.2).to(torch.uint8).to(dev)
        rel = torch.cat([(torch.rand(n, S, K, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        return hip.Seg(kv, 0, 128, T, idx, inv, None, 1, rel=rel)

    segs = [seg(1024, 64), seg(128, 25)]
    out = torch.empty(rows, 640, device=dev)
    flag = torch.empty(rows, dtype=torch.uint8, device=dev)
    buf = (C.c_uint64 * 8)()
    for _ in range(3):
        hip.knarpe_attn(q, 0, 128, bias, n, S, segs, out, flag, fxy, fyw)
    lib.tbx_debug_attn_clock(buf)
    reps = 20
    for _ in range(reps):
        hip.knarpe_attn(q, 0, 128, bias, n, S, segs, out, flag, fxy, fyw)
    lib.tbx_debug_attn_clock(buf)
    passes = buf[4]
    print(f"rows {rows}: {passes / reps:.0f} passes per launch of wave 0; shader cycles per pass:")
    names, vals = (["issue the DMAs of pass p + R - 1", "wait for pass p's DMAs", "LDS reads + the pass's arithmetic"], buf[5:8]) if RING else (PH, buf[:4])
    tot = 0.0
    for name, v in zip(names, vals):
        tot += v / passes
        print(f"    {name:60s} {v / passes:8.0f}")
    print(f"    {'sum':60s} {tot:8.0f}")
