"""GPU parity of the traffic-rule checks: tbx_rule_tables / tbx_rule_check / tbx_rule_accumulate
through the C ABI vs the reference's golden outputs (tests/golden/rules.npz) and vs the oracle (oracle/rule_checks.py).
Boolean results: bit-exact."""
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import rule_checks as R
from test_oracle_rules import EPISODES, run_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(one_hot):
    w = 1 << torch.arange(one_hot.shape[-1], dtype=torch.int32)
    return (one_hot.to(torch.int32) * w).sum(-1).to(torch.uint8)


def _checker(tb, e, rollouts=1):
    T = import_module("trafficbots_amd.utils.traffic_rule_checker")
    d = lambda t: t.to(DEV)
    r = lambda t: d(t.repeat_interleave(rollouts, 0))
    return T.TrafficRuleChecker(mp_boundary=d(e["map/boundary"]), mp_valid=d(e["map/valid"]), mp_type=d(e["map/type"]),
                                mp_pos=d(e["map/pos"]), mp_dir=d(e["map/dir"]), ag_type=r(e["agent/type"]), ag_size=r(e["agent/size"]),
                                ag_goal=None, ag_dest=None, tl_valid=r(e["tl/valid"]), tl_pose=r(e["tl/pose"]), disable_check=False)


def _log_inputs(e, rollouts=1):
    r = lambda t: t.repeat_interleave(rollouts, 0).to(DEV)
    return r(e["agent/valid"]), r(e["agent/pose"]), r(e["agent/motion"]), r(_bits(e["tl/state"]).contiguous())


@pytest.mark.parametrize("tag", sorted(EPISODES))
def test_rule_kernels_vs_reference_golden_and_oracle(tb, golden_dir, tag):
    g = np.load(golden_dir / "rules.npz")
    e = tb.synthetic.make_rule_episode(**EPISODES[tag])
    rc = _checker(tb, e)
    got = rc.check_log(*_log_inputs(e))
    ora, counter = run_oracle(e)
    T = int(g[f"{tag}_n_step"])
    for k in R.RuleCheckOracle.KEYS:
        for s in ("", "_this_step"):
            ref = np.unpackbits(g[f"{tag}_{k}{s}"], axis=-1)[..., :T].astype(bool)
            assert np.array_equal(got[k + s].cpu().numpy(), ref), (tag, k + s, "vs reference golden")
            assert torch.equal(got[k + s].cpu(), ora[k + s]), (tag, k + s, "vs oracle")
    assert torch.equal(rc.passive_counter.cpu(), counter)
    assert np.array_equal(rc.passive_counter.cpu().numpy(), g[f"{tag}_passive_counter"])


def test_per_step_check_equals_log_check_and_resumes(tb):
    """The reference's per-step entry point accumulates like one pass over the log; a log checked in two ranges too."""
    e = tb.synthetic.make_rule_episode(**EPISODES["a"])
    whole = _checker(tb, e).check_log(*_log_inputs(e))
    valid, pose, motion, bits = _log_inputs(e)
    T = valid.shape[2]
    rc = _checker(tb, e)
    tl = e["tl/state"].to(DEV)
    for t in range(T):
        v = rc.check(valid[:, :, t], pose[:, :, t].contiguous(), motion[:, :, t].contiguous(), tl[:, :, t])
        for k, x in v.items():
            assert torch.equal(x, whole[k][:, :, t]), (k, t)
    rc2 = _checker(tb, e)
    first = rc2.check_log(valid, pose, motion, bits, t0=0, n_t=17)
    second = rc2.check_log(valid, pose, motion, bits, t0=17)
    for k in whole:
        assert torch.equal(first[k][:, :, :17], whole[k][:, :, :17]) and torch.equal(second[k][:, :, 17:], whole[k][:, :, 17:]), k


def test_rollouts_share_the_scene_tables(tb):
    """K rollouts of a scene read one copy of the road-edge / lane tables (map_batch_div = K)."""
    e = tb.synthetic.make_rule_episode(**EPISODES["a"])
    one = _checker(tb, e).check_log(*_log_inputs(e))
    K = 3
    many = _checker(tb, e, rollouts=K).check_log(*_log_inputs(e, rollouts=K))
    for k in one:
        assert torch.equal(many[k], one[k].repeat_interleave(K, 0)), k


def test_rule_kernels_vs_oracle_at_scene_size(tb):
    """configs[1] scene size (64 agents, 1024 polylines, 128 lights), a crowded 12-step episode: every flag vs the oracle,
    except where the oracle's own decision is within fp32 round-off of flipping (none expected; reported if any)."""
    e = tb.synthetic.make_rule_episode(n_sc=1, n_ag=64, n_mp=1024, n_tl=128, n_step=12, seed=11, extent=80.0)
    got = _checker(tb, e).check_log(*_log_inputs(e))
    ora, _ = run_oracle(e)
    for k in R.RuleCheckOracle.KEYS:
        assert torch.equal(got[k + "_this_step"].cpu(), ora[k + "_this_step"]), k
        assert torch.equal(got[k].cpu(), ora[k]), k
    assert sum(int(ora[k + "_this_step"].sum()) for k in ("collided", "collided_wosac", "run_road_edge")) > 0


def test_rollout_buffer_carries_rule_violations(tb):
    """WaymoMotion.reactive_replay -> RolloutBuffer.violation holds the five checks for every step, equal to re-checking the
    logged trajectory with the oracle."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    cfg = tb.config.default_model_cfg(n_tgt_knn=4)
    wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(DEV).eval()
    batch = tb.synthetic.make_scene(2, 8, 64, 8, seed=0)
    b = wm.pre_processing({k: v.to(DEV) for k, v in {**batch, **tb.synthetic.to_history_batch(batch)}.items()})
    mp, tl = wm.encode_scene(b, tl_valid_key="gt/tl_valid")
    valid = b["gt/ag_valid"].any(-1)
    buf = wm.reactive_replay(b, mp, tl, torch.zeros(2, 8, 16, device=DEV), valid, b["gt/ag_navi"], valid,
                             wm.teacher_forcing_reactive_replay, True, step_end=20)
    o = R.RuleCheckOracle(b["map/valid"].cpu(), b["map/type"].cpu(), b["map/pos"].cpu(), b["map/dir"].cpu(), b["ref/ag_type"].cpu(),
                          b["ref/ag_size"].cpu(), tl["tl_token_valid"].cpu(), tl["tl_token_pose"].cpu())
    pv, pp, pm, ts = buf.pred_valid[:, 0].cpu(), buf.pred_pose[:, 0].cpu(), buf.pred_motion[:, 0].cpu(), buf.vis_dict["tl_state"][:, 0].cpu()
    for t in range(20):
        v = o.check(pv[:, :, t], pp[:, :, t], pm[:, :, t], ts[:, :, t])
        for k, x in v.items():
            assert torch.equal(buf.violation[k][:, 0, :, t].cpu(), x), (k, t)
