"""Device time of one AdamW step over the default model's 874 parameter tensors: torch's fused multi-tensor form, the foreach form, and the fused
form over ONE flat tensor of the same 10,657,094 elements (what pl_modules/data_parallel.FlatAdamW runs).   python tools/opt_time.py"""
import os, sys
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
dev = torch.device("cuda:0")
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
for mode in ("fused", "foreach"):
    os.environ["TBX_FUSED_ADAMW"] = "1" if mode == "fused" else "0"
    (opt,), _ = wm.configure_optimizers()
    ps = [p for g in opt.param_groups for p in g["params"]]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    print(mode, len(ps), "tensors", sum(p.numel() for p in ps), "params:", e0.elapsed_time(e1) / 10, "ms per step (device, back to back)", opt.defaults.get("fused"))
# one flat tensor of the same size
n = sum(p.numel() for p in ps)
flat = torch.nn.Parameter(torch.zeros(n, device=dev)); flat.grad = torch.randn(n, device=dev) * 1e-3
o = torch.optim.AdamW([flat], lr=1e-4, fused=True)
for _ in range(3): o.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o.step()
e1.record(); torch.cuda.synchronize()
print("one flat tensor, fused:", e0.elapsed_time(e1) / 10, "ms")
