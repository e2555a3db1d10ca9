"""Per-stage cost of tbx_rowchain programs on the GPU (HIP events, 200 repeats each)."""
import sys; sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
load_package()
hip = import_module('trafficbots_amd.hip'); hip.load()
from trafficbots_amd.hip import Chain, BUF0, BUF1, AUX
dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(rows, 640, device=dev); out = torch.empty(rows, 1024, device=dev)
W = {k: torch.randn(*s, device=dev) * 0.05 for k, s in dict(w128=(128, 128), w512=(512, 128), w512b=(128, 512), w384=(384, 128), w896=(896,128)).items()}
b = {k: torch.randn(v.shape[0], device=dev) for k, v in W.items()}
g = torch.ones(128, device=dev); be = torch.zeros(128, device=dev)

def timeit(name, build, ldw=1028, reps=200):
    ch = Chain(16, ldw); build(ch)
    for _ in range(5): ch.run(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ch.run(rows)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:50s} {e0.elapsed_time(e1)/reps*1e3:8.1f} us  ({len(ch.stages)} stages)")

hip.Chain.xcd_hint = False
timeit("load128+store128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.store(BUF0, 0, 128, out)))
timeit("load640+store640", lambda c: (c.load(x, BUF0, 0, n=640), c.store(BUF0, 0, 640, out)))
timeit("+ 1 linear 128->128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w128'], b['w128']), c.store(BUF1, 0, 128, out)))
timeit("+ 2 linear 128->128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w128'], b['w128']), c.linear(BUF1, 0, BUF0, 0, W['w128'], b['w128']), c.store(BUF0, 0, 128, out)))
timeit("+ 4 linear 128->128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), [ (c.linear(BUF0, 0, BUF1, 0, W['w128'], b['w128']), c.linear(BUF1, 0, BUF0, 0, W['w128'], b['w128'])) for _ in range(2)], c.store(BUF0, 0, 128, out)))
timeit("+ 8 linear 128->128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), [ (c.linear(BUF0, 0, BUF1, 0, W['w128'], b['w128']), c.linear(BUF1, 0, BUF0, 0, W['w128'], b['w128'])) for _ in range(4)], c.store(BUF0, 0, 128, out)))
timeit("linear 128->512", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w512'], b['w512']), c.store(BUF1, 0, 512, out)))
timeit("linear 128->512->128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w512'], b['w512']), c.linear(BUF1, 0, BUF0, 0, W['w512b'], b['w512b']), c.store(BUF0, 0, 128, out)))
timeit("linear 128->896", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w896'], b['w896']), c.store(BUF1, 0, 896, out)))
timeit("8x layernorm", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), [c.layernorm(BUF0, 0, BUF0, 0, g, be) for _ in range(8)], c.store(BUF0, 0, 128, out)))
timeit("8x add", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), [c.add(BUF0, 0, BUF1, 0, 128) for _ in range(8)], c.store(BUF0, 0, 128, out)))
timeit("8x rowmask", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), [c.rowmask(BUF0, 0, 128) for _ in range(8)], c.store(BUF0, 0, 128, out)))
timeit("grouped qt 4x(32->128)", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF0, 128, W['w128'], wt=True, groups=4, src_stride=32, dst_stride=128), c.store(BUF0, 128, 512, out)))
timeit("grouped u 4x(128->32)", lambda c: (c.load(x, BUF0, 0, n=640), c.linear(BUF0, 128, BUF0, 0, W['w128'], b['w128'], accum=True, groups=4, src_stride=128, dst_stride=32), c.store(BUF0, 0, 128, out)))
timeit("same small ldw=132: 2 linear 128", lambda c: (c.load(x[:, :128], BUF0, 0, n=128), c.linear(BUF0, 0, BUF1, 0, W['w128'], b['w128']), c.linear(BUF1, 0, BUF0, 0, W['w128'], b['w128']), c.store(BUF0, 0, 128, out)), ldw=132)
