import torch, time
dev = torch.device("cuda:0")
def bench(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (1024, 2048):
    for N, K in ((512, 128), (640, 128), (256, 128), (128, 512), (128, 640), (128, 128), (384, 128)):
        x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        Wt = W.t().contiguous()
        r = {"linear": bench(lambda: torch.nn.functional.linear(x, W, b)),
             "linear_nobias": bench(lambda: torch.nn.functional.linear(x, W)),
             "mm_Wt_contig": bench(lambda: torch.addmm(b, x, Wt)),
             "halves": bench(lambda: torch.cat([torch.nn.functional.linear(x, W[:N // 2], b[:N // 2]), torch.nn.functional.linear(x, W[N // 2:], b[N // 2:])], 1))}
        print(M, N, K, {k: round(v, 1) for k, v in r.items()})
