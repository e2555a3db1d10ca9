#!/usr/bin/env python
"""Per-stage in-kernel cost of live-row (tbx_rowchain_live) LINEAR stages: same weight re-used (L2-hot) vs 20 different
weights, 1 workgroup vs many, warm. Profiling build (make -C trafficbotsv1.5_amd/csrc clk).  usage: live_micro.py [live_rows]"""
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
lib.tbx_debug_sub_dump.argtypes = [C.c_void_p]
from trafficbots_amd.hip import BUF0, BUF1, Chain  # noqa: E402

dev = torch.device("cuda:0")
live = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots = hip.MAX_STAGES + 4


def clocks(ch, rows, flush=None):
    for _ in range(3):
        if flush is not None:
            flush.add_(1.0)
        ch.run(rows)
    torch.cuda.synchronize()
    if flush is not None:
        flush.add_(1.0)
    torch.cuda.synchronize()
    lib.tbx_debug_clock_reset()
    ch.run(rows)
    buf = (C.c_uint64 * (4 * slots))()
    lib.tbx_debug_clock_dump(buf, 4)
    if ch.live_rows and os.environ.get("TBX_SUB"):
        sub = (C.c_uint64 * 8)()
        lib.tbx_debug_sub_dump(sub)
        print("   last gemv stage, shader cycles: entry->barrier %d, ->dma issued %d, ->fma loop done %d, ->epilogue done %d; stage-end %d" % (
            sub[1] - sub[0], sub[2] - sub[1], sub[3] - sub[2], sub[4] - sub[3], buf[len(ch.stages) - 1] and 0))
    return [(buf[j + 1] - buf[j]) / 100.0 for j in range(len(ch.stages))]


for rows in (live, 32 * live, 128 * live):
    x = torch.randn(rows, 128, device=dev)
    out = torch.empty(rows, 128, device=dev)
    for what in ("same", "different", "different+flush"):
        n = 20
        Ws = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(n if what != "same" else 1)]
        bs = [torch.randn(128, device=dev) for _ in Ws]
        flush = torch.zeros(128 << 20, device=dev) if "flush" in what else None  # 512 MB: evicts L2 and the Infinity Cache
        for mode in (live, 0):
            ch = Chain(16, 132, 132, 132, live_rows=mode) if mode else Chain(16, 132, 132, 132)
            ch.load(x, BUF0, 0, n=128)
            for i in range(n):
                src, dst = (BUF0, BUF1) if i % 2 == 0 else (BUF1, BUF0)
                ch.linear(src, 0, dst, 0, Ws[i % len(Ws)], bs[i % len(Ws)], relu=True)
            ch.store(BUF0, 0, 128, out)
            d = clocks(ch, rows, flush)
            lin = d[1:1 + n]
            print(f"rows {rows:4d} grid {(rows + (mode or 16) - 1) // (mode or 16):3d} {'live' + str(mode) if mode else 'mfma16':7s} weights {what:16s}: "
                  f"first {lin[0]:5.2f} us, mean of rest {sum(lin[1:]) / (n - 1):5.2f} us, min {min(lin):5.2f}, total {sum(d):6.1f} us")
        del flush
