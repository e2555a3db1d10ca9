"""Kernel time of WaymoMotion.encode_scene (map encoder + light pre-compute), warm: python tools/scene_encode_profile.py [n_calls]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).eval()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(1, 64, 1024, 128, seed=0).items()}
with torch.no_grad():
    wm.train()  # training-mode pre-processing takes the full-episode keys of a synthetic scene
    b = wm.pre_processing(batch)
    wm.eval()
    for _ in range(3):
        wm.encode_scene(b, tl_valid_key="gt/tl_valid")
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        wm.encode_scene(b, tl_valid_key="gt/tl_valid")
    torch.cuda.synchronize()
    print("encode_scene warm: %.2f ms per call" % ((time.perf_counter() - t0) / n * 1e3))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        wm.encode_scene(b, tl_valid_key="gt/tl_valid")
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
