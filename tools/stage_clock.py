#!/usr/bin/env python
"""In-situ per-stage timing of every row-chain launch of one simulation step (profiling build libtbx_hip_clk.so:
`make -C trafficbotsv1.5_amd/csrc clk`). Workgroup 0 stamps the 100 MHz wall clock at each stage boundary.
    python tools/stage_clock.py [bench.py rollout args]"""
import ctypes as C
import os
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["TBX_HIP_LIB"] = str(ROOT / "trafficbotsv1.5_amd" / "csrc" / "libtbx_hip_clk.so")

import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

OPS = {1: "LOAD", 2: "LINEAR", 3: "LN", 4: "ADD", 5: "COPY", 6: "ROWMASK", 7: "GROUPMAX", 8: "POOLMAX", 9: "STORE", 10: "CLAMP", 11: "DROPOUT", 12: "ATTN", 13: "ATTNSEG"}


def main():
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-graph"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    lib = hip.load()
    lib.tbx_debug_clock_dump.argtypes = [C.c_void_p, C.c_int]
    wm, full = bench.build(tb, args, dev, 0)
    eng, _ = bench.gpu_rollout_setup(tb, wm, full, args, dev)
    eng.run(args.warmup + 3, use_graph=False)
    torch.cuda.synchronize()
    progs = []
    orig = hip.Chain.run

    def run(self, n_rows, group_rows=0):
        progs.append((n_rows, self.tile_rows, [(s.op, s.k, s.n, s.reserved, s.dst) for s in self.stages]))
        return orig(self, n_rows, group_rows)

    hip.Chain.run = run
    lib.tbx_debug_clock_reset()
    eng.run(1, use_graph=False)
    slots = hip.MAX_STAGES + 4
    buf = (C.c_uint64 * (2048 * slots))()
    n = lib.tbx_debug_clock_dump(buf, 2048)
    assert n == len(progs), (n, len(progs))
    tot = 0.0
    for i, (rows, tile, st) in enumerate(progs):
        c = buf[i * slots:(i + 1) * slots]
        grid = c[slots - 1] >> 32
        dur = [(c[j + 1] - c[j]) / 100.0 for j in range(len(st))]
        tot += sum(dur)
        mhz = (c[slots - 2] - c[slots - 3]) / max(sum(dur), 1e-9)
        print(f"launch {i:2d}: rows {rows:6d} tile {tile} grid {grid:4d}  in-kernel {sum(dur):7.1f} us  shader clock {mhz:6.0f} MHz")
        for (op, k, nn, g, dst), d in zip(st, dur):
            extra = f" k={k} n={nn}" + (f" groups={g}" if g else "") + (" ->global" if op == 2 and dst == 3 else "")
            print(f"      {OPS[op]:9s}{extra:34s} {d:7.2f} us")
    print(f"sum of in-kernel chain time (workgroup 0): {tot:.1f} us over {n} launches")


if __name__ == "__main__":
    main()
