"""Micro-repro: column sums (bias gradients) under hipGraph replay."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(rows, n, via):
    x = torch.randn(rows, n, device=dev)
    w = torch.randn(n, 32, device=dev, requires_grad=True)
    b = torch.zeros(n, device=dev, requires_grad=True)
    inp = torch.randn(rows, 32, device=dev)
    def f():
        if via == "sum":
            junk = torch.full((rows,), 316.0, device=dev); del junk
            return x.sum(0)
        y = torch.nn.functional.linear(inp, w, b)
        (g,) = torch.autograd.grad((y * x).sum(), b)
        return g
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = f()
    res = []
    for i in range(3):
        x.copy_(torch.randn(rows, n, device=dev))
        g.replay(); torch.cuda.synchronize()
        ref = x.sum(0)
        res.append(float((out - ref).abs().max()))
    return res
for rows in (128, 1024, 4096, 40960, 131072):
    for n in (64, 128, 384, 512):
        for via in ("sum", "linear"):
            r = run(rows, n, via)
            flag = "BAD" if max(r) > 1e-2 * (rows ** 0.5) else "ok"
            print(rows, n, via, ["%.3g" % v for v in r], flag)
