"""Multi-GPU readiness on ONE GPU (no 8-GPU node exists for the builder): the data-parallel training step
through the real exchange path - a one-rank RCCL process group on the device (torch.distributed backend "nccl" IS RCCL on ROCm):
`broadcast_parameters`, the flat gradient all-reduce of `allreduce_gradients` inside `train_step` and after `GraphedTrainStep`'s
graph replay, per-rank seed separation, and that `bench.py --mode train --gpus 1` reports what the default line's `training`
entry reports. No scaling curve is measured here (DESIGN.md 7 says so)."""
import json
import os
import subprocess
import sys
from importlib import import_module
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import json, os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
cfg = tb.config.default_model_cfg(n_tgt_knn=4)
scfg = tb.config.default_sim_cfg()
scfg["time_step_end"] = 20
out = {}
torch.manual_seed(99)  # a rank whose RNG is NOT rank 0's: the broadcast makes the weights rank 0's anyway
wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **scfg).to(dev).train()
before = DP.parameters_checksum(wm.model)
out["broadcast_bytes"] = DP.broadcast_parameters(wm.model)
out["checksum_kept"] = bool(torch.equal(before, DP.parameters_checksum(wm.model)))  # one rank: src == self
chk = DP.parameters_checksum(wm.model).clone()
dist.all_reduce(chk, op=dist.ReduceOp.MAX)
out["checksum_equal_across_ranks"] = bool(torch.equal(chk, DP.parameters_checksum(wm.model)))
out["rank_seeds"] = [DP.rank_seed(1234, r) for r in range(8)]
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(2, 8, 64, 8, seed=0).items()}
# eager step: fwd + bwd + flat all-reduce (RCCL) + clip + AdamW
calls = {"n": 0, "bytes": 0}
real = DP.allreduce_gradients
def counted(params, *a, **k):
    b = real(params, *a, **k)
    calls["n"] += 1
    calls["bytes"] += b
    return b
DP.allreduce_gradients = counted
m = DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
live = DP.live_parameters(wm.model)
out["eager_loss_finite"] = bool(torch.isfinite(m["loss"]))
out["eager_allreduce_bytes"] = calls["bytes"]
out["live_bytes"] = sum(p.numel() for p in live) * 4
print("eager done", flush=True)
del m, live
# graphed step: replay, then the same exchange outside the graph
calls["n"] = calls["bytes"] = 0
gs = DP.GraphedTrainStep(wm, opt, batch, warmup=1, verbose=True)
for _ in range(2):
    m = gs(batch)
print("graph steps done", flush=True)
out["graph_loss_finite"] = bool(torch.isfinite(m["loss"]))
out["graph_allreduce_calls"] = calls["n"]
out["graph_allreduce_bytes"] = calls["bytes"]
out["graph_live_bytes"] = sum(p.numel() for p in gs.live) * 4
out["graph_flat_bytes"], out["graph_n_live"] = gs.flat.nbytes, len(gs.live)
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
'''


def _run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=str(ROOT), env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_one_rank_rccl_training_step_takes_the_real_exchange_path(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    out = _run([sys.executable, str(script), str(ROOT)], env)
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["broadcast_bytes"] > 40e6  # 10.66 M parameters + buffers as fp32
    assert res["checksum_kept"] and res["checksum_equal_across_ranks"]
    assert len(set(res["rank_seeds"])) == 8 and res["rank_seeds"][0] == 1234
    assert res["eager_loss_finite"] and res["graph_loss_finite"]
    assert res["eager_allreduce_bytes"] == res["live_bytes"] > 30e6  # every live gradient travelled, as ONE flat buffer
    # (the captured step's buffer starts every parameter's slice at a multiple of 64 elements - FlatAdamW lays the parameters' data out
    #  the same way: the exchange carries the gaps, < 256 bytes per parameter)
    assert res["graph_allreduce_calls"] == 2 and res["graph_allreduce_bytes"] == 2 * res["graph_flat_bytes"]
    assert 0 <= res["graph_flat_bytes"] - res["graph_live_bytes"] < 256 * res["graph_n_live"]
    assert res["graph_live_bytes"] == res["live_bytes"]


def test_bench_train_mode_agrees_with_the_default_lines_training_entry():
    """`bench.py --mode train --gpus 1` and the `training` entry the default line appends run the same step: same metric, config
    and all-reduce size; throughput within the run-to-run spread of a 2-step measurement."""
    a = json.loads(_run([sys.executable, "bench.py", "--mode", "train", "--gpus", "1", "--scenes", "2", "--steps", "2", "--warmup", "1",
                         "--agents", "32", "--polylines", "128", "--lights", "32"]).strip().splitlines()[-1])
    assert a["metric"] == "training scenes/sec" and a["n_gpus"] == 1 and a["finite"] and a["steps"] == 2
    assert a["config"]["parallelism"] == "dp1" and a["config"]["global_batch"] == 2
    assert a["config"]["allreduce_bytes"] > 30e6
    assert abs(a["value"] - 2 * 2 / (a["ms_per_step"] * 2e-3)) < 2e-5 * a["value"]  # (the line rounds to 6 significant digits)


def test_two_ranks_through_bench_on_one_gpu():
    """The N > 1 flow of bench.py end to end on a one-GPU box: `--gpus 2` starts its own two ranks (tools/benchlib/launch.py), both on
    cuda:0 over gloo (TBX_BENCH_SHARE_GPU / TBX_BENCH_BACKEND: RCCL refuses two ranks on one device) - rendezvous, scene sharding,
    barriers, the MAX-over-ranks clock, rank 0's line, the flat gradient all-reduce of the training step. A functional check of the
    multi-rank path (run.py:50-52, strategy="ddp"); the numbers of two processes sharing a GPU mean nothing."""
    env = {"TBX_BENCH_SHARE_GPU": "1", "TBX_BENCH_BACKEND": "gloo"}
    small = ["--agents", "32", "--polylines", "128", "--lights", "32"]
    a = json.loads(_run([sys.executable, "bench.py", "--gpus", "2", "--steps", "8", "--warmup", "2", "--no-wosac-shape", "--no-bf16-shape",
                         "--no-train-shape", "--profile-steps", "0", "--new-scenes", "0", *small], env).strip().splitlines()[-1])
    assert a["n_gpus"] == 2 and a["finite"] and a["steps"] == 8 and a["scaling"] == "weak" and "cpu_baseline" not in a
    assert abs(a["value"] - 2 * 32 * 8 / (a["ms_per_step"] * 8e-3)) < 2e-5 * a["value"]  # both ranks' agent-steps over the slower rank's time
    b = json.loads(_run([sys.executable, "bench.py", "--mode", "train", "--gpus", "2", "--scenes", "2", "--steps", "2", "--warmup", "1",
                         *small], env).strip().splitlines()[-1])
    assert b["metric"] == "training scenes/sec" and b["n_gpus"] == 2 and b["finite"]
    assert b["config"]["parallelism"] == "dp2" and b["config"]["global_batch"] == 4 and b["config"]["allreduce_bytes"] > 30e6


def test_loss_terms_with_empty_counters_are_left_out(tb):
    """A per-GPU batch without a single valid traffic light (plausible on WOMD at batch 3): the reference leaves a term whose
    counter is zero out of the loss (metrics/training.py:166-186); the step must not turn NaN."""
    dev = torch.device("cuda:0")
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    scfg = tb.config.default_sim_cfg()
    scfg["time_step_end"] = 14
    torch.manual_seed(0)
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **scfg).to(dev).train()
    batch = tb.synthetic.make_scene(2, 8, 64, 8, seed=3)
    for k in list(batch):
        if k in ("tl_lane/valid", "tl_stop/valid", "tl_lane/state", "tl_stop/state"):
            batch[k] = torch.zeros_like(batch[k])
    loss = wm.training_step({k: v.to(dev) for k, v in batch.items()}, 0)
    loss.backward()
    m = wm.last_metrics
    assert bool(torch.isfinite(loss)) and float(m["tl_state_loss"]) == 0.0
    assert abs(float(m["loss"]) - float(m["vae_kl"] - m["diffbar_reward"] + m["navi_loss"])) < 1e-5 * max(1.0, abs(float(m["loss"])))


@pytest.mark.gpu
def test_flat_adamw_equals_the_fused_optimizer_and_keeps_its_state_layout(tb):
    """pl_modules/data_parallel.FlatAdamW (parameters, moments and step counters as slices of flat buffers, ONE fused launch per run of a
    parameter group) against torch.optim.AdamW(fused=True) stepping the same parameters with the same gradients: bit-identical
    parameters and moments after several steps with two parameter groups (different lr) and an lr change in between; the optimizer's own
    state entries stay per-parameter tensors (state_dict round trip), and a direct optimizer.step() afterwards continues from the same state."""
    dev = torch.device("cuda:0")
    DP = import_module("trafficbots_amd.pl_modules.data_parallel")
    g = torch.Generator().manual_seed(0)
    shapes = [(128, 128), (128,), (640, 128), (5,), (256, 128), (1, 128), (121, 31)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).to(dev)) for i, s in enumerate(shapes)]
        opt = torch.optim.AdamW(ps[:4], lr=1e-3, weight_decay=0.01, fused=True)
        opt.add_param_group({"params": ps[4:], "lr": 3e-4})
        return ps, opt

    (pa, oa), (pb, ob) = make(), make()
    grads = [[(torch.randn(s, generator=g) * 0.1).to(dev) for s in shapes] for _ in range(6)]
    flat = DP.FlatGrads(pa)
    assert DP.FlatAdamW.usable(oa, flat)
    fa = DP.FlatAdamW(oa, flat)
    assert len(fa.runs) == 2 and all(p.data_ptr() != 0 for p in pa)
    for t, gs in enumerate(grads[:4]):
        if t == 2:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-4  # (what an lr scheduler does)
        for p, q, gr in zip(pa, pb, gs):
            q.grad = gr.clone()
        torch._foreach_copy_(flat.views, gs)
        fa.step()
        ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p, q)
        assert torch.equal(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"]) and torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"])
        assert float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == 4.0
    sd = oa.state_dict()
    assert len(sd["state"]) == len(shapes) and tuple(sd["state"][6]["exp_avg"].shape) == shapes[6]
    # ... and the optimizer's own step() goes on from there (the state entries are ordinary per-parameter tensors: views of the flat buffers)
    for gs in grads[4:]:
        for p, q, gr in zip(pa, pb, gs):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p, q) and float(oa.state[p]["step"]) == 6.0
