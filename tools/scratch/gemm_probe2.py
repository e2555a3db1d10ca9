import torch
dev = torch.device("cuda:0")
shapes = [(1024,512,128,11),(2048,512,128,13),(1024,640,128,17),(2048,640,128,19),(1024,256,128,23),(2048,256,128,29),(16384,512,128,31),(16384,640,128,37),(16384,256,128,41),(1024,128,512,43),(2048,128,512,47),(1024,128,640,53),(45056,64,128,59),(22528,64,128,61)]
for M,N,K,c in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for _ in range(c): torch.nn.functional.linear(x, W, b)
torch.cuda.synchronize()
print("shapes:", shapes)
