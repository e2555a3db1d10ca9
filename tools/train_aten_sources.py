"""Where do the torch (aten) launches of ONE eager training step of configs[2] come from?  A TorchDispatchMode logs every aten op
that reaches the device with its output size and the innermost frame of this package on the Python stack (ops run by the autograd
engine itself - gradient accumulation - have no such frame: they are listed under the backward node that was running), grouped by
(op, source): calls and output megabytes. The step's own HIP launches (C ABI through ctypes) do not pass the dispatcher: this is the
list of what is NOT yet one of them.   python tools/train_aten_sources.py [bf16|fp32] [top]"""
import os, sys, traceback
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from torch.utils._python_dispatch import TorchDispatchMode
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
wm.train_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
step = lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
step(); step()
torch.cuda.synchronize()

VIEW = ("view", "reshape", "expand", "permute", "transpose", "select", "slice", "unsqueeze", "squeeze", "as_strided", "detach", "alias", "t.default",
        "unbind", "split", "chunk", "_unsafe_view", "empty", "size", "stride", "is_", "_local_scalar", "item", "lift_fresh", "unflatten", "narrow", "unfold",
        "_reshape_alias", "view_as", "resize_", "set_", "record_stream", "_has_compatible", "prim.", "sym_")
rec = {}
PKG = os.sep + "trafficbotsv1.5_amd" + os.sep


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW):
            return out
        t = out if torch.is_tensor(out) else next((o for o in (out if isinstance(out, (tuple, list)) else ()) if torch.is_tensor(o)), None)
        if t is None or not t.is_cuda:
            t = next((a for a in args if torch.is_tensor(a) and a.is_cuda), None)
            if t is None:
                return out
        node = torch._C._current_autograd_node() if hasattr(torch._C, "_current_autograd_node") else None
        src = "(autograd engine)" if node is None else f"(engine: {node.name()})"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if PKG in fr.filename:
                src = f"{fr.filename.split(PKG)[-1]}:{fr.lineno} {fr.name}"
                break
        k = (name.replace("aten.", ""), src, tuple(t.shape) if t.dim() <= 4 else (t.numel(),))
        e = rec.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += t.numel() * t.element_size() / 1e6
        return out


with Log():
    step()
torch.cuda.synchronize()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tot = sum(v[0] for v in rec.values())
print(f"{tot} device aten ops in one eager step ({wm.train_precision} class); by source, most calls first")
by_src = {}
for (op, src, shp), (n, mb) in rec.items():
    e = by_src.setdefault((op, src), [0, 0.0, set()])
    e[0] += n
    e[1] += mb
    e[2].add(shp)
for (op, src), (n, mb, shp) in sorted(by_src.items(), key=lambda kv: -kv[1][0])[:top]:
    ex = sorted(shp, key=lambda s: -len(s))[:2]
    print(f"{n:5d} x {op:34s} {mb:9.1f} MB  {src:64s} {ex}")
print("\nby (op, source, shape), most output bytes first:")
for (op, src, shp), (n, mb) in sorted(rec.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:5d} x {op:34s} {mb:9.1f} MB  {src:56s} {shp}")
print("\nby op:")
by_op = {}
for (op, src, shp), (n, mb) in rec.items():
    e = by_op.setdefault(op, [0, 0.0])
    e[0] += n
    e[1] += mb
for op, (n, mb) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{n:5d} x {op:34s} {mb:9.1f} MB")
