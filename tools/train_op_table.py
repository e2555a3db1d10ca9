"""Which aten operators (with input shapes) the GPU time of one eager training step (configs[2]) outside this repo's kernels
goes to, forward and backward separately. Usage: python tools/train_op_table.py [rows]"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
torch.cuda.synchronize()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 70
from torch.profiler import profile, ProfilerActivity


def table(fn, title):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
        r = fn()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    dt = lambda e: getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    tot = sum(dt(e) for e in ka)
    print(f"== {title}: {tot / 1e3:.1f} ms of GPU time in {sum(e.count for e in ka if dt(e) > 0)} operator calls")
    for e in sorted(ka, key=dt, reverse=True)[:rows]:
        if dt(e) <= 0:
            break
        print(f"{dt(e) / 1e3:8.2f} ms {e.count:6d} x  {e.key[:44]:44s} {str(e.input_shapes)[:150]}")
    return r


loss = table(lambda: wm.training_step({k: v.clone() for k, v in batch.items()}, 0), "forward (stepping pass + batched forward + loss)")
table(lambda: loss.backward(), "backward")
