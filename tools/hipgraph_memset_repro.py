"""Repro of the ROCm 7.0 hipGraph hazard the captured training step has to avoid (pl_modules/data_parallel.GraphedTrainStep):
a captured chain whose inputs change between replays; with `sum` (torch's multi-block reduction zeroes its semaphores with a
memset node) replays after the first return stale results unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is in the environment.

    python tools/hipgraph_memset_repro.py 3000 sum            # BAD on the default AQL-packet path
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/hipgraph_memset_repro.py 3000 sum clone slice churn   # ok
    flags: sum | clone (memcpy node) | slice (zeros + slice copy) | kcopy (strided copy kernel) | churn (small-block reuse)
"""
import sys, threading, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
flags = set(sys.argv[2:])
rows, n = 4096, 64
x = torch.randn(rows, n, device=dev)
ws = torch.randn(64, n, device=dev)
def f():
    acc = torch.zeros(n, device=dev)
    for i in range(N_IT):
        y = x * ws[i % 64]
        b = y.sum(0) if "sum" in flags else y[i % rows] * 2.0
        if "churn" in flags:
            t1 = torch.full((n,), 316.0, device=dev); acc = acc + 0.0 * t1
        if "clone" in flags:
            b = b.clone()                     # contiguous D2D copy -> memcpy node
        if "slice" in flags:
            z = torch.zeros(3 * n, device=dev); z[:n].copy_(b); b = z[:n]
        if "kcopy" in flags:
            z = torch.empty(n, 2, device=dev); z[:, 0].copy_(b); b = z[:, 0]   # strided copy -> kernel
        acc = acc + b
    return acc
s = torch.cuda.Stream()
res = {}
def cap():
    torch.cuda.set_device(dev)
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        res["acc"] = f()
    res["g"] = g
old = threading.stack_size(1 << 30)
t = threading.Thread(target=cap); t.start(); t.join()
threading.stack_size(old)
errs = []
for r in range(4):
    x.copy_(torch.randn(rows, n, device=dev))          # inputs change between replays
    res["g"].replay(); torch.cuda.synchronize()
    ref = f()
    errs.append(float((res["acc"] - ref).abs().max() / ref.abs().max()))
print(sorted(flags), "rel err per replay", ["%.2g" % e for e in errs], "BAD" if max(errs) > 1e-3 else "ok")
