#!/usr/bin/env python
"""Phase timing inside tbx_layer_tile (profiling build libtbx_hip_clk.so: `make -C trafficbotsv1.5_amd/csrc clk`): thread 0 of workgroup
0 stamps the shader clock at the phase boundaries of every tile_layer_kernel launch of one eager simulation step.
    python tools/tile_clock.py --agents 128 --rollouts 32 [bench.py rollout args]"""
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

PH = ["launch -> x / attention rows in LDS (planes written, barrier)", "value fold (1 unit)", "out_proj + residual (1 unit)", "LayerNorm 2 -> planes",
      "linear1 + relu (4 units)", "linear2 + residual + store x (4 units)", "LayerNorm 3 -> planes", "q | k | v (1 / 3 units) + stores", "W_k^T q (1 unit) + stores drained"]


def main():
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-graph"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    lib = hip.load()
    lib.tbx_debug_tl_dump.argtypes = [C.c_void_p, C.c_int]
    wm, full = bench.build(tb, args, dev, 0)
    eng, _ = bench.gpu_rollout_setup(tb, wm, full, args, dev)
    eng.sched = eng.sched.replace(lights_ahead=False)  # (one stream: nothing runs beside the launch that is timed)
    eng.run(args.warmup + 3, use_graph=False)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (256 * 16))()
    lib.tbx_debug_tl_dump(buf, 256)  # reset
    eng.run(1, use_graph=False)
    torch.cuda.synchronize()
    n = lib.tbx_debug_tl_dump(buf, 256)
    for i in range(n):
        c = buf[i * 16:(i + 1) * 16]
        kind = int(c[15])
        print(f"launch {i}: tile_layer_kernel<ATTN={kind // 100}, FFN={kind // 10 % 10}, PROJ={kind % 10}>  {(c[9] - c[0]) / 100.0:7.2f} x 100 shader clocks in workgroup 0")
        last = c[0]
        for j in range(1, 10):
            if c[j] > last:
                print(f"    {PH[j - 1]:70s} {(c[j] - last) / 100.0:6.2f}")
                last = c[j]
        if c[14] > c[10] > 0:  # linear1's first unit taken apart (profiling build only)
            print(f"    [linear1 unit 0: wait for its weights {(c[11] - c[10]) / 100.0:.2f}; load latency of the next unit {(c[12] - c[11]) / 100.0:.2f}; "
                  f"LDS reads + 12 MFMAs + sum {(c[13] - c[12]) / 100.0:.2f}; relu / split / planes write {(c[14] - c[13]) / 100.0:.2f}]")


if __name__ == "__main__":
    main()
