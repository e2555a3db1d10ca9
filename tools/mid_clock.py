#!/usr/bin/env python
"""Phase timing inside tbx_knarpe_dec_mid (profiling build libtbx_hip_clk.so: `make -C trafficbotsv1.5_amd/csrc clk`): workgroup 0
stamps the shader clock at the phase boundaries of every launch of one eager simulation step.
    python tools/mid_clock.py [bench.py rollout args]"""
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

PH = ["prologue (DMA requests, x, LN params, q loads)", "self sweep + slot merge", "combine + value fold",
      "out_proj GEMV (+ W_q request)", "LayerNorm (+ W_kf request)", "wait for W_q", "q GEMV", "W_k^T q GEMV",
      "cross sweep + slot merge", "combine + value fold + out_proj2", "LayerNorm 2", "linear1 + relu", "linear2 + store x",
      "LayerNorm 3 + next q | k | v", "next W_k^T q"]
# the all-matrix-path kernel (Schedule.dec_tail_mfma, the default): dec_layer_mf_kernel's stamps
PH_MF = ["prologue (weight units 0, 1 requested, x, LN params, q loads)", "self sweep + slot merge", "combine + value fold",
         "out_proj", "LayerNorm 1", "q", "W_k^T q", "q / qt loads of the cross sweep", "cross sweep + slot merge",
         "combine + value fold + out_proj2", "LayerNorm 2", "linear1 + relu", "linear2 + store x", "LayerNorm 3 + next q | k | v",
         "next W_k^T q"]


def main():
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-graph"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tb = load_package()
    hip = import_module("trafficbots_amd.hip")
    lib = hip.load()
    lib.tbx_debug_mid_dump.argtypes = [C.c_void_p, C.c_int]
    wm, full = bench.build(tb, args, dev, 0)
    eng, _ = bench.gpu_rollout_setup(tb, wm, full, args, dev)
    if not os.environ.get("TBX_CLOCK_TWO_STREAM"):  # (default: the one-stream order; TBX_CLOCK_TWO_STREAM=1: the timed schedule with the fused step tail)
        eng.sched = eng.sched.replace(lights_ahead=False)
    eng.run(args.warmup + 3, use_graph=False)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (256 * 16))()
    lib.tbx_debug_mid_dump(buf, 256)  # reset
    eng.run(1, use_graph=False)
    torch.cuda.synchronize()
    n = lib.tbx_debug_mid_dump(buf, 256)
    for i in range(n):
        c = buf[i * 16:(i + 1) * 16]
        d = [(c[j + 1] - c[j]) / 100.0 if c[j + 1] > c[j] else 0.0 for j in range(15)]  # clock64 = s_memtime = shader clocks here (~2.4 GHz: 100 clocks = 0.042 us)
        print(f"launch {i}: {sum(d):6.2f} x 100 shader clocks in workgroup 0")
        if c[14] > c[13] and c[15] > c[14] and c[0] > c[15]:  # the agents' last layer with the fused step tail: stamp 0 was rewritten at its end
            print(f"    [heads: {(c[14] - c[13]) / 100.0:.2f}; tbx_sim_step of the row's agent: {(c[15] - c[14]) / 100.0:.2f}; next tbx_agent_prep: {(c[0] - c[15]) / 100.0:.2f}]")
        for name, v in zip(PH if os.environ.get("TBX_DEC_TAIL_MFMA") == "0" else PH_MF, d):
            print(f"    {name:70s} {v:6.2f}")


if __name__ == "__main__":
    main()
