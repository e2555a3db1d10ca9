"""N > 1 path on CPU (gloo, world_size 2): scenes are sharded across ranks with NO data-path collective; the only
collectives bench.py issues are the barrier and the MAX-reduce of the wall time. This test runs the same sharding and
reduction logic (on the oracle as the per-rank worker, since there is no GPU here) and checks that the union of the
ranks' rollouts equals the single-process result."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package

    tb = load_package()
    bench = __import__("bench")
    torch.set_num_threads(2)
    scenes = bench.shard_scenes(n_total=4, rank=rank, world=world)
    res = {s: _rollout_checksum(tb, s) for s in scenes}
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the wall-time reduction of bench.py
    assert float(t) == float(world)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def _rollout_checksum(tb, seed):
    from oracle import trafficbots_oracle as O
    from importlib import import_module

    cfg = tb.config.default_model_cfg(n_tgt_knn=4)
    model = import_module("trafficbots_amd.models.traffic_bots").TrafficBots(**cfg)
    tb.utils.det_fill(model, 0)
    P = {k: v.detach() for k, v in model.state_dict().items()}
    batch = tb.synthetic.make_scene(1, 8, 64, 8, seed=seed)
    b = O.scene_centric({**batch, **tb.synthetic.to_history_batch(batch)}, training=False)
    om = O.TrafficBotsOracle(P, cfg)
    with torch.no_grad():
        mp_ = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_)
        v = b["sc/ag_valid"].any(-1)
        ro = O.Sim(om, tb.config.default_sim_cfg(), False).rollout(
            b, mp_, tl, torch.zeros(1, 8, 16), v, b["gt/ag_navi"], v, tb.config.default_sim_cfg().teacher_forcing_joint_future_pred, 3)
    return float(ro["pred_pose"].double().sum())


def test_two_rank_sharding_equals_single_process(tmp_path, tb):
    world, port = 2, 29533
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = {}
    for r in range(world):
        got.update(torch.load(tmp_path / f"r{r}.pt"))
    assert sorted(got) == [0, 1, 2, 3]  # every scene simulated exactly once
    ref = {s: _rollout_checksum(tb, s) for s in (0, 3)}
    for s, v in ref.items():
        assert abs(got[s] - v) < 1e-9 * max(1.0, abs(v))


# ---------------------------------------------------------------------------------------------------- DDP wiring
def _ddp_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module

    from __graft_entry__ import load_package

    load_package()
    dp = import_module("trafficbots_amd.pl_modules.data_parallel")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    dead = torch.nn.Linear(3, 3)  # never used: must be skipped, as the 2.75 M gradient-less parameters are
    x = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    xs = x[rank * 2:(rank + 1) * 2]  # each rank: its shard of the global batch
    net(xs).pow(2).mean().backward()
    live = dp.live_parameters(torch.nn.ModuleList([net, dead]))
    assert len(live) == 4
    nbytes = dp.allreduce_gradients(live)
    assert nbytes == sum(p.numel() for p in live) * 4
    torch.save([p.grad.clone() for p in live], os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_allreduce_equals_full_batch(tmp_path):
    world, port = 2, 29541
    mp.spawn(_ddp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    x = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    net(x).pow(2).mean().backward()
    for a, b, p in zip(g0, g1, net.parameters()):
        assert torch.equal(a, b)  # every rank ends with the same averaged gradient
        torch.testing.assert_close(a, p.grad, rtol=1e-6, atol=1e-7)  # = gradient of the full batch


def _flat_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module

    from __graft_entry__ import load_package

    load_package()
    dp = import_module("trafficbots_amd.pl_modules.data_parallel")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    x = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    xs = x[rank * 2:(rank + 1) * 2]
    net(xs).pow(2).mean().backward()  # first backward: finds the live parameters
    fg = dp.FlatGrads(dp.live_parameters(net))
    for _ in range(2):  # two further steps: backward accumulates straight into the flat buffer, which is what travels
        fg.zero()
        fg.attach()
        net(xs).pow(2).mean().backward()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(fg.params, fg.views))  # still views: nothing was re-allocated
        nbytes = dp.allreduce_gradients(fg)
    assert nbytes == fg.nbytes == sum(p.numel() for p in net.parameters()) * 4
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f"f{rank}.pt"))
    dist.destroy_process_group()


def test_flat_gradient_buffer_allreduce_equals_full_batch(tmp_path):
    """FlatGrads (the live gradients as views of one persistent buffer, all-reduced in place): 2 ranks x half the batch each ==
    the full batch's gradient, on every rank, with no gather / scatter copies."""
    world, port = 2, 29547
    mp.spawn(_flat_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = torch.load(tmp_path / "f0.pt"), torch.load(tmp_path / "f1.pt")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    x = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    net(x).pow(2).mean().backward()
    for a, b, p in zip(g0, g1, net.parameters()):
        assert torch.equal(a, b)
        torch.testing.assert_close(a, p.grad, rtol=1e-6, atol=1e-7)
