"""tbx_agent_prep outputs of one library build on a seeded input -> gpurun_out/prep_<tag>.pt (compare two builds bit by bit)."""
import os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
tb = load_package()
from trafficbots_amd import hip
tag = sys.argv[1]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n, A, W, M = 2, 64, 11, 128
hv = (torch.rand(n, A, W, generator=g) < 0.8).to(torch.uint8).to(dev)
hp = ((torch.rand(n, A, W, 3, generator=g) - 0.5) * torch.tensor([300., 300., 6.28])).to(dev)
hm = torch.randn(n, A, W, 3, generator=g).to(dev)
attr6 = torch.rand(n, A, 6, generator=g).to(dev)
ty = torch.randint(0, 3, (n, A), generator=g).to(torch.uint8).to(dev)
fxy = torch.repeat_interleave(torch.exp(torch.linspace(0, 3, 8)) * 0.01, 2).to(dev)
fyaw = torch.repeat_interleave(torch.arange(1, 17).float(), 2).to(dev)
dest = torch.randint(0, M, (n, A), generator=g).to(dev)
mpp = ((torch.rand(n, M, 3, generator=g) - 0.5) * 300).to(dev)
out = dict(tok_pose=torch.zeros(n * A, 3, device=dev), tok_invalid=torch.zeros(n * A, dtype=torch.uint8, device=dev),
           attr=torch.zeros(n * A * W, 32, device=dev), pe=torch.zeros(n * A * W, 64, device=dev), row_invalid=torch.zeros(n * A * W, dtype=torch.uint8, device=dev),
           type_mask=torch.zeros(3, n * A, dtype=torch.uint8, device=dev), navi_pose3=torch.zeros(n * A, 3, device=dev), navi_row=torch.zeros(n * A, dtype=torch.int32, device=dev))
hip.agent_prep(hv, hp, hm, attr6, ty, fxy, fyaw, 64, out, dest=dest, mp_tok_pose=mpp, n_mp=M, mp_batch_div=1)
torch.cuda.synchronize()
os.makedirs(ROOT / "gpurun_out", exist_ok=True)
torch.save({k: v.cpu() for k, v in out.items()}, ROOT / "gpurun_out" / f"prep_{tag}.pt")
