"""GPU parity of the boundary rows the round-1 review found open (VERDICT r01 "Next round" 1-2):
  * `WaymoMotion.forward` with the reference's arguments, `rollout(..., player_policy)`, `hparams` (waymo_motion.py:118-311);
  * the inference `RolloutBuffer` fields `tl_state_nll`, `diffbar_reward`, `action_log_prob`, `mask_teacher_forcing`
    (buffer.py:39-146) vs the oracle AND the reference's golden values;
  * `NaviPredictor.forward` on the HIP chain kernel vs the reference's golden log-probabilities (navigation.py:175-278);
  * the WOSAC shape (32 rollouts x 128 agents, shared map + shared lights) vs the oracle, rollout by rollout;
  * 80 free-running closed-loop steps at the 64-agent / 1024-polyline / 128-light scene (damped action head) vs the oracle.
"""
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import rule_checks as R
from oracle import trafficbots_oracle as O
from test_hip_rollout import _compare, _oracle_tokens, _setup

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _damp(wm, P, f=0.02):
    with torch.no_grad():
        for k, p in wm.model.state_dict().items():
            if k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k:
                p.mul_(f)
                P[k] = P[k] * f


def _replay_args(tb, wm, b, bd, om, mp_o, tl_o):
    with torch.no_grad():
        post_o = om.latent_encoder(b["gt/ag_valid"], b["sc/ag_attr"], b["gt/ag_motion"], b["gt/ag_pose"], b["ref/ag_type"],
                                   b["gt/tl_state"], mp_o, tl_o, posterior=True)
    return post_o.mean, post_o.valid, b["gt/ag_valid"].any(-1)


def test_hparams_are_the_constructor_arguments(tb):
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    scfg = tb.config.default_sim_cfg()
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **scfg)
    for k in ("time_step_current", "time_step_end", "p_training_rollout_prior", "training_detach_model_input", "lr_navi"):
        assert wm.hparams[k] == scfg[k] and getattr(wm.hparams, k) == scfg[k]
    assert wm.hparams.differentiable_reward.l_rot.weight == scfg["differentiable_reward"]["l_rot"]["weight"]


def test_rollout_buffer_fields_vs_oracle_and_reference(tb, golden_dir):
    """tl_state_nll / diffbar_reward / mask_teacher_forcing / action_log_prob of a reactive replay (the reference fills them at
    waymo_motion.py:250-300; tbx_sim_step logs them per step) vs the oracle's Sim.rollout and the reference's golden buffer."""
    dev = torch.device(DEV)
    wm, P, b, bd = _setup(tb, dev, (8, 64, 8), 4)
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=4), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    mp_o, tl_o = _oracle_tokens(om, b)
    z, zv, navi_v = _replay_args(tb, wm, b, bd, om, mp_o, tl_o)
    n_roll, n_cmp = 90, 16
    with torch.no_grad():
        ro = O.Sim(om, scfg, False).rollout(b, mp_o, tl_o, z, zv, b["gt/ag_navi"], navi_v, scfg.teacher_forcing_joint_future_pred, n_roll)
    mp, tl = wm.encode_scene(bd, tl_valid_key="gt/tl_valid")
    buf = wm.reactive_replay(bd, mp, tl, z.to(dev), zv.to(dev), bd["gt/ag_navi"], navi_v.to(dev), wm.teacher_forcing_joint_future_pred,
                             True, step_end=n_roll)
    sl = slice(0, n_cmp)
    # the light recurrence reads no agent: teacher-forced for the whole ground truth, so its NLL compares over all 90 steps
    assert torch.equal(buf.tl_state_nll_invalid[:, 0].cpu(), ro["tl_state_nll_invalid"])
    torch.testing.assert_close(buf.tl_state_nll[:, 0].cpu(), ro["tl_state_nll"], rtol=1e-4, atol=2e-5)
    assert torch.equal(buf.mask_teacher_forcing[:, 0].cpu(), ro["mask_teacher_forcing"])
    assert torch.equal(buf.diffbar_reward["diffbar_reward_valid"][:, 0, :, sl].cpu(), ro["diffbar_reward_valid"][:, :, sl])
    # the reward is -(0.1 SmoothL1 + 10 * cosine + ..) of a pose that agrees to ~1e-4 over these steps
    torch.testing.assert_close(buf.diffbar_reward["diffbar_reward"][:, 0, :, sl].cpu(), ro["diffbar_reward"][:, :, sl], rtol=1e-3, atol=2e-4)
    r = buf.diffbar_reward
    torch.testing.assert_close(r["diffbar_reward"], (r["r_imitation_pos"] + r["r_imitation_rot"]) + r["r_imitation_spd"], rtol=0, atol=0)
    # action_log_prob: log N(mean | mean, exp(log_std)) of the 2-d action, 0 for invalid agents (dynamics.py:87-91)
    log_std = torch.stack([P[f"action_head.log_std.{i}"] for i in range(3)], 0)  # [3, 2]
    lp_type = -(log_std.sum(-1) + 2 * 0.5 * np.log(2 * np.pi))
    lp = (b["ref/ag_type"].float() * lp_type).sum(-1).unsqueeze(-1) * ro["pred_valid"].float()
    torch.testing.assert_close(buf.action_log_prob[:, 0, :, sl].cpu(), lp[:, :, sl], rtol=1e-6, atol=1e-6)
    assert buf.navi_log_prob.shape == (1, 1, 8, 1) and torch.equal(buf.navi_log_prob_valid[:, 0, :, 0].cpu(), navi_v)
    buf.compute_log_prob(None)
    assert buf.log_prob.shape == (1, 1, 8)
    # ... and the REFERENCE's own buffer
    g = np.load(golden_dir / "model_c1.npz")
    np.testing.assert_allclose(buf.tl_state_nll[:, 0].cpu().numpy(), g["rr_tl_state_nll"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(buf.diffbar_reward["diffbar_reward"][:, 0, :, sl].cpu().numpy(), g["rr_diffbar_reward"][:, :, sl],
                               rtol=1e-3, atol=5e-4)


def test_forward_with_reference_arguments_equals_reactive_replay(tb):
    """A step-wise driver written against the reference - teacher_forcing.get -> WaymoMotion.forward(mp_tokens, tl_tokens,
    ag_override, tl_override, player_override, deterministic_action) -> rule check -> buffer.add -> dynamics.disable_ag /
    disable_navi (waymo_motion.py:232-275) - must produce the engine's rollout bit for bit: same kernels, same inputs."""
    dev = torch.device(DEV)
    wm, P, b, bd = _setup(tb, dev, (8, 64, 8), 4)
    mp, tl = wm.encode_scene(bd, tl_valid_key="gt/tl_valid")
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 8, 16, generator=g).to(dev)
    valid = bd["gt/ag_valid"].any(-1)
    T = 40
    fast = wm.reactive_replay(bd, mp, tl, z, valid, bd["gt/ag_navi"], valid, wm.teacher_forcing_joint_future_pred, True, step_end=T)
    ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["gt/ag_valid"],
                 "gt_pose": bd["gt/ag_pose"], "gt_motion": bd["gt/ag_motion"], "ag_latent": z, "ag_latent_valid": valid,
                 "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid}
    # (1) the packaged loop
    slow = wm.rollout(ag_tokens, mp, tl, bd["gt/tl_state"], wm.teacher_forcing_joint_future_pred, wm._rule_checker(bd, bd["gt/ag_navi"], tl),
                      T, True, stepwise=True)
    slow.flatten_joint_future(1)
    # (2) the same loop written out here, the way a user of the reference would (20 steps are enough to cover overrides + free steps)
    tf = wm.teacher_forcing_joint_future_pred
    eng = wm.begin_rollout(ag_tokens, mp, tl, bd["gt/tl_state"], tf, wm._rule_checker(bd, bd["gt/ag_navi"], tl), T, stepwise=True)
    dyn, poses, valids = wm.dynamics, [], []
    for step in range(1, 21):
        ag_override, tl_override = tf.get(step, dyn.ag_valid, dyn.ag_pose, dyn.ag_motion)
        pred, vis = wm.forward(mp_tokens=mp, tl_tokens=tl, ag_override=ag_override, tl_override=tl_override, player_override=None,
                               deterministic_action=True)
        assert set(pred) == {"action_log_prob", "pred_valid", "pred_pose", "pred_motion", "pred_tl_state_dist"}
        assert set(vis) == {"pred_valid", "pred_pose", "pred_motion", "action", "ag_navi", "ag_navi_valid", "navi_reached", "tl_state"}
        poses.append(pred["pred_pose"]), valids.append(pred["pred_valid"])
        viol = {"outside_map_this_step": eng.S["now_outside"].bool(), "dest_reached_this_step": eng.S["now_reached"].bool()}
        dyn.disable_ag(viol, bd["gt/ag_valid"][:, :, step])
        dyn.disable_navi(viol)
    assert torch.equal(torch.stack(poses, 2), fast.pred_pose[:, 0, :, :20])
    assert torch.equal(torch.stack(valids, 2), fast.pred_valid[:, 0, :, :20])
    for name in ("pred_valid", "pred_pose", "pred_motion", "tl_state_nll", "tl_state_nll_invalid", "action_log_prob",
                 "mask_teacher_forcing", "navi_log_prob", "navi_log_prob_valid"):
        assert torch.equal(getattr(slow, name), getattr(fast, name)), name
    for k in fast.violation:
        assert torch.equal(slow.violation[k], fast.violation[k]), k
    for k in fast.diffbar_reward:
        assert torch.equal(slow.diffbar_reward[k], fast.diffbar_reward[k]), k
    assert torch.equal(slow.vis_dict["action"], fast.vis_dict["action"])
    assert torch.equal(slow.vis_dict["tl_state"], fast.vis_dict["tl_state"])


def test_player_policy_overrides_actions(tb):
    """rollout(..., player_policy): the player's physical action replaces the policy's for the agents it claims (dynamics.py:
    104-107). A player that hands back the policy's own logged actions reproduces the rollout bit for bit; a constant-action
    player drives its agent along the closed-form MultiPath++ trajectory while the others' first free step is unchanged."""
    dev = torch.device(DEV)
    wm, P, b, bd = _setup(tb, dev, (8, 64, 8), 4, ragged=False)
    mp, tl = wm.encode_scene(bd, tl_valid_key="gt/tl_valid")
    valid = bd["gt/ag_valid"].any(-1)
    z = torch.zeros(1, 8, 16, device=dev)
    T = 25
    ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["gt/ag_valid"],
                 "gt_pose": bd["gt/ag_pose"], "gt_motion": bd["gt/ag_motion"], "ag_latent": z, "ag_latent_valid": valid,
                 "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid}
    run = lambda pol: wm.rollout(ag_tokens, mp, tl, bd["gt/tl_state"], wm.teacher_forcing_joint_future_pred,
                                 wm._rule_checker(bd, bd["gt/ag_navi"], tl), T, True, player_policy=pol)
    base = run(lambda pose: None)
    state = {"t": 0}

    def echo(pose):  # the policy's own action of this step, claimed for every agent
        t = state["t"]
        state["t"] += 1
        return {"valid": torch.ones(1, 8, dtype=torch.bool, device=dev), "action": base.vis_dict["action"][:, :, t]}

    same = run(echo)
    assert torch.equal(same.pred_pose, base.pred_pose) and torch.equal(same.vis_dict["action"], base.vis_dict["action"])

    def constant(pose):
        v = torch.zeros(1, 8, dtype=torch.bool, device=dev)
        v[0, 0] = True
        a = torch.zeros(1, 8, 2, device=dev)
        a[0, 0, 0], a[0, 0, 1] = 1.0, 0.1
        return {"valid": v, "action": a}

    drv = run(constant)
    act = drv.vis_dict["action"]
    pv = drv.pred_valid[0, 0]
    assert torch.equal(act[0, 0][pv], torch.tensor([1.0, 0.1], device=dev).expand(int(pv.sum()), 2))
    first_free = 10  # log slot of step 11: the first step after the warm start whose state the player touched is slot 11
    assert torch.equal(drv.pred_pose[0, 1:, first_free], base.pred_pose[0, 1:, first_free])
    # agent 0, free-running from slot 10 on (teacher forcing ends at step 10): closed form of dynamics.py:237-274
    p, m = drv.pred_pose[0, 0], drv.pred_motion[0, 0]
    for t in range(first_free + 1, T):
        if not bool(pv[t]) or not bool(pv[t - 1]):
            continue
        v_t, th_t = m[t - 1, 0] + 0.05 * 1.0, p[t - 1, 2] + 0.05 * 0.1
        exp = torch.stack([p[t - 1, 0] + 0.1 * v_t * torch.cos(th_t), p[t - 1, 1] + 0.1 * v_t * torch.sin(th_t), p[t - 1, 2] + 0.1 * 0.1])
        torch.testing.assert_close(p[t], exp, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("sizes,knn,tag", [((8, 64, 8), 4, "c1"), ((64, 1024, 128), 32, "c2")])
def test_navi_predictor_forward_vs_reference_golden(tb, golden_dir, sizes, knn, tag):
    """NaviPredictor.forward (navigation.py:175-278) on the chain kernel - destination logits of every (agent, polyline) pair -
    vs the log-probabilities / argmax / validity the REFERENCE produced on the same seeded scene, and vs the oracle."""
    dev = torch.device(DEV)
    wm, P, b, bd = _setup(tb, dev, sizes, knn)
    om = O.TrafficBotsOracle(P, tb.config.default_model_cfg(n_tgt_knn=knn), training=False)
    mp_o, _ = _oracle_tokens(om, b)
    mp = wm.model.mp_encoder(bd["sc/mp_valid"], bd["sc/mp_attr"], bd["sc/mp_pose"], bd["ref/mp_type"])
    pred = wm.model.navi_predictor(ag_valid=bd["sc/ag_valid"], ag_attr=bd["sc/ag_attr"], ag_motion=bd["sc/ag_motion"],
                                   ag_pose=bd["sc/ag_pose"], ag_type=bd["ref/ag_type"], **mp)
    g = np.load(golden_dir / f"model_{tag}.npz")
    assert np.array_equal(pred.valid.cpu().numpy(), g["navi_valid"])
    lp = pred.log_prob(bd["gt/ag_navi"]).cpu().numpy()
    ok = g["navi_valid"] & np.isfinite(g["navi_log_prob_gt"])
    assert np.array_equal(np.isfinite(lp)[g["navi_valid"]], np.isfinite(g["navi_log_prob_gt"])[g["navi_valid"]])
    np.testing.assert_allclose(lp[ok], g["navi_log_prob_gt"][ok], rtol=2e-3, atol=2e-3)
    with torch.no_grad():
        po = om.navi_predictor(b["sc/ag_valid"], b["sc/ag_attr"], b["sc/ag_motion"], b["sc/ag_pose"], b["ref/ag_type"], mp_o)
    probs, probs_o = pred.probs.cpu(), po.distribution.probs
    torch.testing.assert_close(probs, probs_o, rtol=5e-3, atol=2e-5)
    # the argmax agrees wherever the reference's top two probabilities are separated by more than the tolerance
    top2 = probs_o.topk(2, -1)[0]
    clear = torch.from_numpy(g["navi_valid"]) & ((top2[..., 0] - top2[..., 1]) > 1e-3 * top2[..., 0])
    assert np.array_equal(probs.argmax(-1)[clear].numpy(), g["navi_argmax"][clear.numpy()])
    assert int(clear.sum()) >= int(0.5 * g["navi_valid"].sum())


# Tolerances of the WOSAC-shape check per schedule: (pose / action atol over the 16 compared steps of three rollouts, pose / action atol over
# the 10 warm-start steps of all 32, light-state NLL atol). "reduced" = Schedule.reduced(), the bf16-ARITHMETIC schedule (bf16 tables,
# tbx_knarpe_attn_fwd_mfma, tbx_layer_tile_bf16 / tbx_heads_tile_bf16 / tbx_window_tile_bf16 for the agents' 4096 rows, dec_layer_mf1_kernel
# for the lights' 128): 2 x the largest difference measured against the ORACLE on MI355X (the test prints them;
# profiles/r05_reduced_tolerances.txt). Validity, light states, map-exit flags and the rule flags stay bit-identical in both.
# (the free steps: the random-weight loop amplifies a difference ~2-3x per free step - DESIGN.md 2 -, so the bf16-arithmetic schedule is held
# point-wise over the warm start + 3 free steps (n_cmp 13) and the fp32-class default over 6 (n_cmp 16))
# measured (max |d pose| / |d action| over the first 10 / 13 / 16 steps of three rollouts): default 1.4e-5 / 1.4e-4, 4.4e-5 / 1.9e-4,
# 2.0e-4 / 1.1e-3, |d nll| 1.0e-4; reduced 8.0e-3 / 8.0e-2, 1.75e-2 / 0.117, 5.5e-2 / 0.51, |d nll| 4.8e-2 (actions reach 7)
WOSAC_TOL = {"default": dict(n_cmp=16, free=(4e-4, 2.3e-3), warm=(4e-5, 4e-4), nll=2e-4),
             "reduced": dict(n_cmp=13, free=(0.035, 0.24), warm=(0.016, 0.16), nll=0.1)}


def test_wosac_shape_joint_futures_vs_oracle(tb):
    """BASELINE config 5 at its own size: 32 rollouts x 128 agents / 1024 polylines / 128 lights through
    `joint_future_pred` - map tokens and K/V tables shared by the 32 rollouts (batch_div), lights stepped once per scene
    (share_lights) - vs the oracle's Sim.rollout run rollout by rollout with THAT rollout's sampled latent and destination.
    First 16 steps (10 warm start + 6 free; the loop is chaotic after that); rule flags bit-exact vs the oracle's
    checks on the logged trajectories. Both schedules (default, Schedule.reduced(): WOSAC_TOL) against the same oracle runs: the
    sampled latents / destinations are those of the default run (same generator seed: asserted equal)."""
    dev = torch.device(DEV)
    K, A, T = 32, 128, 16
    SUB = list(range(0, K, 4))
    wm, P, b, bd = _setup(tb, dev, (A, 1024, 128), 32)
    E = import_module("trafficbots_amd.engine")
    D = import_module("trafficbots_amd.models.modules.distributions")
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=32), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    dmax = lambda x, y: float((x - y).abs().max())
    valid = bd["sc/ag_valid"].any(-1)
    onehot = torch.nn.functional.one_hot(bd["gt/ag_navi"], 1024).float()
    wm.hp.joint_future_pred_deterministic_k0 = False
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    sim = O.Sim(om, scfg, False)
    vc = valid.cpu()
    oracle = {}  # (the oracle's rollouts: made once, from the first schedule's sampled latents / destinations)
    for sched in ("default", "reduced"):
        tol = WOSAC_TOL[sched]
        wm.schedule = E.DEFAULT if sched == "default" else E.DEFAULT.reduced()
        wm.engine_cache = 0
        mp, tl = wm.encode_scene(bd, n_rollout=K)
        lat = D.DiagGaussian(torch.zeros(1, A, 16, device=dev), torch.zeros(16, device=dev), valid=valid)  # std-normal prior
        torch.manual_seed(21)
        buf = wm.joint_future_pred(bd, mp, tl, lat, D.DestCategorical(probs=onehot, valid=valid), wm.teacher_forcing_joint_future_pred, K,
                                   step_end=T)
        eng = wm._engine
        assert eng.tl_div == K and eng.n == K  # the benchmarked sharing is what ran
        z_all = eng.ag_latent.view(K, A, 16).cpu()
        dest_all = eng.dest.cpu()
        assert float((z_all[0] - z_all[1]).abs().max()) > 0.1
        if not oracle:
            oracle["z"], oracle["dest"] = z_all, dest_all
            for k in (0, 13, 31):
                with torch.no_grad():
                    oracle[k] = sim.rollout(bh, mp_o, tl_o, z_all[k:k + 1], vc, dest_all[k:k + 1], vc, scfg.teacher_forcing_joint_future_pred, T,
                                            gt_prefix="hist", tl_gt_key="sc/tl_state")
            # EVERY 4TH of the 32 rollouts over the 10 warm-start steps (poses are teacher-forced there, so the oracle runs them as one batch
            # of 8 scene copies, each with its rollout's latent and destination - all 32 cost the CPU oracle two minutes): action means and
            # light states of rollouts 0, 4, .., 28 beside the three full rollouts above
            rep = lambda t: t.repeat_interleave(len(SUB), 0) if torch.is_tensor(t) and t.shape[0] == 1 else t
            bK = {k: rep(v) for k, v in bh.items()}
            mpK = {k: rep(v) for k, v in mp_o.items()}
            tlK = {k: rep(v) for k, v in tl_o.items()}
            with torch.no_grad():
                oracle["K"] = sim.rollout(bK, mpK, tlK, z_all[SUB], vc.expand(len(SUB), -1), dest_all[SUB], vc.expand(len(SUB), -1),
                                          scfg.teacher_forcing_joint_future_pred, 10, gt_prefix="hist", tl_gt_key="sc/tl_state")
        else:  # the same samples under both schedules (the generator, not the arithmetic, draws them)
            assert torch.equal(z_all, oracle["z"]) and torch.equal(dest_all, oracle["dest"])
        meas = {"nll": 0.0}
        for k in (0, 13, 31):
            ro = oracle[k]
            for h in (10, 13, 16):  # what the test measured, per horizon (printed below)
                hs = slice(0, h)
                meas[f"pose{h}"] = max(meas.get(f"pose{h}", 0.0), dmax(buf.pred_pose[:, k, :, hs].cpu(), ro["pred_pose"][:, :, hs]))
                meas[f"action{h}"] = max(meas.get(f"action{h}", 0.0), dmax(buf.vis_dict["action"][:, k, :, hs].cpu(), ro["action"][:, :, hs]))
            meas["nll"] = max(meas["nll"], dmax(buf.tl_state_nll[:, k, :, :16].cpu(), ro["tl_state_nll"][:, :, :16]))
        print(f"[wosac shape vs oracle, {sched}] 3 rollouts, max |d pose| / |d action| over the first 10 / 13 / 16 steps: "
              + " ; ".join(f"{meas[f'pose{h}']:.3g} / {meas[f'action{h}']:.3g}" for h in (10, 13, 16)) + f"; |d nll| {meas['nll']:.3g}")
        n_cmp = tol["n_cmp"]
        for k in (0, 13, 31):
            ro = oracle[k]
            sl = slice(0, n_cmp)
            assert torch.equal(buf.pred_valid[:, k, :, sl].cpu(), ro["pred_valid"][:, :, sl]), k
            assert torch.equal(buf.vis_dict["tl_state"][:, k, :, sl].cpu(), ro["tl_state"][:, :, sl]), k
            assert torch.equal(buf.violation["outside_map"][:, k, :, sl].cpu(), ro["outside_map"][:, :, sl]), k
            torch.testing.assert_close(buf.pred_pose[:, k, :, sl].cpu(), ro["pred_pose"][:, :, sl], rtol=1e-4, atol=tol["free"][0])
            torch.testing.assert_close(buf.vis_dict["action"][:, k, :, sl].cpu(), ro["action"][:, :, sl], rtol=1e-3, atol=tol["free"][1])
            torch.testing.assert_close(buf.tl_state_nll[:, k, :, sl].cpu(), ro["tl_state_nll"][:, :, sl], rtol=1e-3, atol=tol["nll"])
        roK = oracle["K"]
        sl = slice(0, 10)
        assert torch.equal(buf.pred_valid[0, SUB, :, sl].cpu(), roK["pred_valid"][:, :, sl])
        assert torch.equal(buf.vis_dict["tl_state"][0, SUB, :, sl].cpu(), roK["tl_state"][:, :, sl])
        print(f"[wosac shape vs oracle, {sched}] 10 warm-start steps of 8 of the 32 rollouts: |d pose| {dmax(buf.pred_pose[0, SUB, :, sl].cpu(), roK['pred_pose'][:, :, sl]):.3g}, "
              f"|d action| {dmax(buf.vis_dict['action'][0, SUB, :, sl].cpu(), roK['action'][:, :, sl]):.3g}")
        torch.testing.assert_close(buf.pred_pose[0, SUB, :, sl].cpu(), roK["pred_pose"][:, :, sl], rtol=1e-4, atol=tol["warm"][0])
        torch.testing.assert_close(buf.vis_dict["action"][0, SUB, :, sl].cpu(), roK["action"][:, :, sl], rtol=1e-3, atol=tol["warm"][1])
        assert torch.isfinite(buf.pred_pose).all() and torch.isfinite(buf.vis_dict["action"]).all()  # (all 32)
        assert float((roK["action"][0] - roK["action"][1]).abs().max()) > 1e-3  # the rollouts' policies do differ (their latents do)
        # rule flags of three rollouts, bit-exact against the oracle's checks on the logged trajectories
        ks = [0, 13, 31]
        r = lambda t: t.repeat_interleave(len(ks), 0).cpu()
        o = R.RuleCheckOracle(r(b["map/valid"]), r(b["map/type"]), r(b["map/pos"]), r(b["map/dir"]), r(b["ref/ag_type"]), r(b["ref/ag_size"]),
                              tl["tl_token_valid"][ks].cpu(), tl["tl_token_pose"][ks].cpu())
        pick = lambda t: t[0, ks].cpu()
        pv, pp, pm, ts = pick(buf.pred_valid), pick(buf.pred_pose), pick(buf.pred_motion), pick(buf.vis_dict["tl_state"])
        for t in range(T):
            v = o.check(pv[:, :, t], pp[:, :, t], pm[:, :, t], ts[:, :, t])
            for key, x in v.items():
                assert torch.equal(pick(buf.violation[key])[:, :, t], x), (key, t)


def test_c2_free_rollout_80_steps_damped_policy(tb):
    """BASELINE config 2 over its whole horizon: 10 teacher-forced + 80 free-running closed-loop steps of the 64-agent /
    1024-polyline / 128-light scene, hipGraph replay, with the action head's output layer scaled by 0.02 (the random-weight
    loop is chaotic otherwise) vs the oracle, point-wise."""
    dev = torch.device(DEV)
    wm, P, b, bd = _setup(tb, dev, (64, 1024, 128), 32)
    _damp(wm, P)
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=32), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 64, 16, generator=g)
    valid = b["sc/ag_valid"].any(-1)
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    T = 90
    with torch.no_grad():
        ro = O.Sim(om, scfg, False).rollout(bh, mp_o, tl_o, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, T,
                                            gt_prefix="hist", tl_gt_key="sc/tl_state")
    mp, tl = wm.encode_scene(bd)
    ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["sc/ag_valid"],
                 "gt_pose": bd["sc/ag_pose"], "gt_motion": bd["sc/ag_motion"], "ag_latent": z.to(dev), "ag_latent_valid": valid.to(dev),
                 "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid.to(dev)}
    buf = wm.rollout(ag_tokens, mp, tl, bd["sc/tl_state"], wm.teacher_forcing_joint_future_pred,
                     wm._rule_checker(bd, bd["gt/ag_navi"], tl), T, True)
    buf.flatten_joint_future(1)
    _compare(buf, ro, T, 5e-3)
    torch.testing.assert_close(buf.tl_state_nll[:, 0].cpu(), ro["tl_state_nll"], rtol=1e-3, atol=1e-4)
    # ---- the same loop under the bf16-arithmetic schedule (Schedule.reduced()) against the SAME oracle run: with the damped head the loop
    # does not amplify, so it is compared point-wise over all 90 steps - validity, flags and light states identical, poses / motion /
    # actions within <= 2 x measured (0.184 m-or-rad / 0.0215 / 0.0194 on MI355X; profiles/r05_reduced_tolerances.txt)
    wm.schedule = import_module("trafficbots_amd.engine").DEFAULT.reduced()
    wm.engine_cache = 0
    mp, tl = wm.encode_scene(bd)
    buf = wm.rollout(ag_tokens, mp, tl, bd["sc/tl_state"], wm.teacher_forcing_joint_future_pred,
                     wm._rule_checker(bd, bd["gt/ag_navi"], tl), T, True)
    buf.flatten_joint_future(1)
    dmax = lambda x, y: float((x.cpu() - y).abs().max())
    print(f"[C2 damped 90-step loop, reduced vs oracle] max |d pose| {dmax(buf.pred_pose[:, 0], ro['pred_pose']):.3g}, |d motion| "
          f"{dmax(buf.pred_motion[:, 0], ro['pred_motion']):.3g}, |d action| {dmax(buf.vis_dict['action'][:, 0], ro['action']):.3g}")
    assert torch.equal(buf.pred_valid[:, 0].cpu(), ro["pred_valid"]) and torch.equal(buf.vis_dict["tl_state"][:, 0].cpu(), ro["tl_state"])
    assert torch.equal(buf.violation["outside_map"][:, 0].cpu(), ro["outside_map"]) and torch.equal(buf.violation["dest_reached"][:, 0].cpu(), ro["dest_reached"])
    assert dmax(buf.pred_pose[:, 0], ro["pred_pose"]) <= 0.37 and dmax(buf.pred_motion[:, 0], ro["pred_motion"]) <= 0.045
    assert dmax(buf.vis_dict["action"][:, 0], ro["action"]) <= 0.04


@pytest.mark.parametrize("sizes,knn,K", [((8, 64, 8), 4, 1), ((16, 64, 8), 4, 4)])
def test_scene_loader_graph_equals_eager_refill(tb, sizes, knn, K):
    """pl_modules/scene_loader.SceneLoader: [encoders + RolloutEngine.refill + prime] captured as ONE hipGraph on static inputs. Three
    different scenes loaded through the graph give, bit for bit, the rollout logs of an engine refilled eagerly with the same scenes
    (which test_cached_engine_refilled_in_place_equals_fresh_engines ties to fresh engines) - poses, validity, actions, light states,
    rewards - and the device-side light-sharing check of a refill passes (K > 1)."""
    from types import SimpleNamespace

    from tools.benchlib import rollout as R

    dev = torch.device("cuda:0")
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    SL = import_module("trafficbots_amd.pl_modules.scene_loader")
    Eng = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine
    E = import_module("trafficbots_amd.engine")
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=knn), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(dev).eval()
    wm.schedule = E.DEFAULT.replace(graph_steps=4)
    a = SimpleNamespace(rollouts=K, scenes=1, agents=sizes[0])
    T = 24

    def scene(seed):
        batch = tb.synthetic.make_scene(1, *sizes, seed=seed)
        full = {**batch, **tb.synthetic.to_history_batch(batch)}
        return wm.pre_processing({k: v.to(dev) for k, v in full.items()})

    scenes = [scene(s) for s in (21, 22, 23)]
    keys = ("out_pose", "out_valid", "out_motion", "out_action", "out_tl_state", "out_reward", "out_reward_valid", "out_tf", "out_tl_nll",
            "out_outside_map", "out_dest_reached")
    logs = {}
    # "overlap": the serving loop's order (scene_loader.py's docstring, bench.py's end_to_end_value) - scene k + 1's [encoders + derived
    # state] replayed on the side stream WHILE scene k's step graphs run, its log read after that prefetch was enqueued
    for mode in ("eager", "graph", "overlap"):
        with E.use(wm.schedule):
            eng = Eng(wm.model, wm.dynamics, dev, schedule=wm.schedule)
            eng.reset(**R.engine_inputs(wm, scenes[0], a, dev, T))
            eng.capture()
            loader = SL.SceneLoader(eng, scenes[0], lambda sb: R.engine_inputs(wm, sb, a, dev, T)) if mode != "eager" else None
            out = []
            order = scenes[1:] + scenes[:1]
            if mode == "overlap":
                loader.prefetch(order[0])
            for i, bd in enumerate(order):
                if mode == "overlap":
                    loader.commit()
                    if i + 1 < len(order):
                        loader.prefetch(order[i + 1])
                elif loader is not None:
                    loader.load(bd)
                else:
                    eng.refill(**R.engine_inputs(wm, bd, a, dev, T))
                eng.run(T, use_graph=True)
                if mode != "overlap":
                    torch.cuda.synchronize()
                out.append({k: eng.S[k].clone() for k in keys})  # (stream-ordered behind the rollout)
                eng.buffer(10)  # (reads the device-side light-sharing flag of the refill: must not raise)
            torch.cuda.synchronize()
        logs[mode] = out
    assert not torch.equal(logs["eager"][0]["out_pose"], logs["eager"][1]["out_pose"])  # the scenes do differ
    for mode in ("graph", "overlap"):
        for i, (e, g) in enumerate(zip(logs["eager"], logs[mode])):
            for k in keys:
                assert torch.equal(e[k], g[k]), (mode, i, k)


def test_light_sharing_flag_belongs_to_the_committed_scene(tb):
    """in the overlapped order commit(k); prefetch(k + 1); run(); buffer() the light-sharing flag read by buffer()
    must be scene k's - an engine-owned copy made by graph_commit - not the tensor in graph_prepare's pool that prefetch(k + 1)
    rewrites. Scenes: A (its K rollouts share their lights), B (rollout 1's light states differ: must be refused), C (shared again).
    buffer() passes for A although B's prepare has already run, raises for B although C's has, and passes for C."""
    from types import SimpleNamespace

    from tools.benchlib import rollout as R

    dev = torch.device("cuda:0")
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    SL = import_module("trafficbots_amd.pl_modules.scene_loader")
    Eng = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine
    E = import_module("trafficbots_amd.engine")
    sizes, K, T = (16, 64, 8), 4, 14
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(dev).eval()
    wm.schedule = E.DEFAULT.replace(graph_steps=4)
    a = SimpleNamespace(rollouts=K, scenes=1, agents=sizes[0])

    def scene(seed, mismatch):
        batch = tb.synthetic.make_scene(1, *sizes, seed=seed)
        full = {**batch, **tb.synthetic.to_history_batch(batch)}
        bd = wm.pre_processing({k: v.to(dev) for k, v in full.items()})
        bd["test/mismatch"] = torch.full((1,), float(mismatch), device=dev)
        return bd

    def make_kw(sb):
        kw = R.engine_inputs(wm, sb, a, dev, T)
        g = kw["tl_state_gt"]  # [K, L, T, 5]: rollout 1's ground-truth light states shifted by one state where the scene says so
        bad = g.clone()
        bad[1] = g[1].roll(1, -1)
        kw["tl_state_gt"] = torch.where(sb["test/mismatch"].view(1, 1, 1, 1) > 0, bad, g)
        return kw

    A, B, C = scene(31, 0), scene(32, 1), scene(33, 0)
    with E.use(wm.schedule):
        eng = Eng(wm.model, wm.dynamics, dev, schedule=wm.schedule)
        eng.reset(**make_kw(A))
        assert eng.tl_div == K
        eng.capture()
        loader = SL.SceneLoader(eng, A, make_kw)
        loader.prefetch(A)
        loader.commit()
        loader.prefetch(B)
        eng.run(T, use_graph=True)
        eng.buffer(10)  # A: fine, although B's prepare has already been replayed
        loader.commit()
        loader.prefetch(C)
        eng.run(T, use_graph=True)
        with pytest.raises(RuntimeError, match="lights"):
            eng.buffer(10)  # B: refused, although C's prepare has already been replayed
        loader.commit()
        eng.run(T, use_graph=True)
        eng.buffer(10)  # C
        torch.cuda.synchronize()


def test_submission_shape_128_joint_futures_rule_checks_and_filter(tb):
    """The reference's WOSAC SUBMISSION shape (configs/resume/submission.yaml:5: n_joint_future 128; configs/model/sim_agent.yaml:12's 32
    is the training-time validation value): 128 joint futures x 128 agents / 1024 polylines / 128 lights = 16,384 agent rows in one
    engine, map and light tables shared by the 128 rollouts. Three rollouts against the oracle over the warm start + the first free
    steps (each with ITS sampled latent and destination), every one of the 128 finite and distinct, then the consumer chain end to end:
    rule flags over the whole log -> `_filter_futures` keeps the 32 least-violating futures (scores bit-exact against the oracle's,
    kept set = the 32 smallest scores, trajectories gathered from the log)."""
    from oracle import wosac_filter as F

    dev = torch.device(DEV)
    K, A, T, n_cmp = 128, 128, 14, 14
    wm, P, b, bd = _setup(tb, dev, (A, 1024, 128), 32)
    D = import_module("trafficbots_amd.models.modules.distributions")
    PP = import_module("trafficbots_amd.data_modules.wosac_post_processing")
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=32), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    mp, tl = wm.encode_scene(bd, n_rollout=K)
    valid = bd["sc/ag_valid"].any(-1)
    lat = D.DiagGaussian(torch.zeros(1, A, 16, device=dev), torch.zeros(16, device=dev), valid=valid)  # std-normal prior
    onehot = torch.nn.functional.one_hot(bd["gt/ag_navi"], 1024).float()
    wm.hp.joint_future_pred_deterministic_k0 = False
    torch.manual_seed(5)
    buf = wm.joint_future_pred(bd, mp, tl, lat, D.DestCategorical(probs=onehot, valid=valid), wm.teacher_forcing_joint_future_pred, K, step_end=T)
    eng = wm._engine
    assert eng.tl_div == K and eng.n == K and buf.pred_pose.shape == (1, K, A, T, 3)
    assert torch.isfinite(buf.pred_pose).all() and torch.isfinite(buf.pred_motion).all() and torch.isfinite(buf.vis_dict["action"]).all()
    free = buf.pred_pose[0, :, :, -1]  # the K futures are K different futures
    assert float((free[1:] - free[:-1]).abs().amax(dim=(1, 2)).min()) > 1e-4
    z_all, dest_all = eng.ag_latent.view(K, A, 16).cpu(), eng.dest.cpu()
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    sim, vc = O.Sim(om, scfg, False), valid.cpu()
    for k in (0, 77, 127):
        with torch.no_grad():
            ro = sim.rollout(bh, mp_o, tl_o, z_all[k:k + 1], vc, dest_all[k:k + 1], vc, scfg.teacher_forcing_joint_future_pred, T,
                             gt_prefix="hist", tl_gt_key="sc/tl_state")
        sl = slice(0, n_cmp)
        assert torch.equal(buf.pred_valid[:, k, :, sl].cpu(), ro["pred_valid"][:, :, sl]), k
        assert torch.equal(buf.vis_dict["tl_state"][:, k, :, sl].cpu(), ro["tl_state"][:, :, sl]), k
        torch.testing.assert_close(buf.pred_pose[:, k, :, sl].cpu(), ro["pred_pose"][:, :, sl], rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(buf.vis_dict["action"][:, k, :, sl].cpu(), ro["action"][:, :, sl], rtol=1e-3, atol=2e-3)
    # consumer chain: flags of the whole log (checked against the oracle at the WOSAC shape above and in test_hip_rules.py) -> filter
    post = PP.WOSACPostProcessing(step_gt=90, step_current=10, const_vel_z_sim=True, const_vel_no_sim=True, w_road_edge=0.5, use_wosac_col=True)
    trajs = post._filter_futures(buf, bd["ref/ag_role"])
    assert trajs.shape == (1, 32, A, T - 10, 3)
    score = F.rollout_scores(buf.violation["collided_wosac"].cpu(), buf.violation["run_road_edge"].cpu(), b["ref/ag_role"], 10, 0.5)
    assert torch.equal(post.last_score.cpu(), score)
    idx = post.last_idx.cpu().long()
    assert torch.equal(score[0, idx[0]].sort()[0], score[0].sort()[0][:32]) and len(set(idx[0].tolist())) == 32
    assert torch.equal(trajs.cpu(), buf.pred_pose.cpu()[:, idx[0]][:, :, :, 10:])


def _buffers_equal(a, b, what=""):
    for name in ("pred_valid", "pred_pose", "pred_motion", "tl_state_nll", "tl_state_nll_invalid", "action_log_prob", "mask_teacher_forcing",
                 "navi_log_prob", "navi_log_prob_valid"):
        assert torch.equal(getattr(a, name), getattr(b, name)), (what, name)
    for k in b.violation:
        assert torch.equal(a.violation[k], b.violation[k]), (what, k)
    for k in b.diffbar_reward:
        assert torch.equal(a.diffbar_reward[k], b.diffbar_reward[k]), (what, k)
    assert torch.equal(a.vis_dict["action"], b.vis_dict["action"]) and torch.equal(a.vis_dict["tl_state"], b.vis_dict["tl_state"]), what


@pytest.mark.parametrize("sizes,knn", [((8, 64, 8), 4), ((128, 1024, 128), 32)])
def test_validation_step_in_the_references_calling_order(tb, sizes, knn):
    """The reference's `validation_step` (waymo_motion.py:526-600) replayed statement by statement against this package's modules:
    pre_processing -> model.mp_encoder -> model.tl_encoder.pre_compute(**mp_tokens) -> latent posterior / prior -> navi_predictor ->
    reactive_replay -> joint_future_pred(n_joint_future = hparams.n_joint_future_wosac = 32), every call with the reference's
    keyword arguments and nothing else (no encode_scene, no n_rollout, no step_end). `joint_future_pred` repeats the light tokens
    itself (the reference's :458-462) - on a copy: the caller's dicts are unchanged - and shares the map tokens / K/V tables and
    the lights across the 32 rollouts; results are BIT-IDENTICAL to the path that every oracle comparison of this suite drives
    (`encode_scene(n_rollout=K)`: test_wosac_shape_joint_futures_vs_oracle, test_submission_shape_...)."""
    dev = torch.device(DEV)
    wm, P, b, batch = _setup(tb, dev, sizes, knn)
    model, K = wm.model, wm.hparams.n_joint_future_wosac
    assert K == 32

    def encode_and_predict(mp_tokens, tl_tokens):
        latent_post = model.latent_encoder(ag_valid=batch["gt/ag_valid"], ag_attr=batch["sc/ag_attr"], ag_motion=batch["gt/ag_motion"],
                                           ag_pose=batch["gt/ag_pose"], ag_type=batch["ref/ag_type"], tl_state=batch["gt/tl_state"],
                                           mp_tokens=mp_tokens, tl_tokens=tl_tokens, posterior=True)
        latent_prior = model.latent_encoder(ag_valid=batch["sc/ag_valid"], ag_attr=batch["sc/ag_attr"], ag_motion=batch["sc/ag_motion"],
                                            ag_pose=batch["sc/ag_pose"], ag_type=batch["ref/ag_type"], tl_state=batch["sc/tl_state"],
                                            mp_tokens=mp_tokens, tl_tokens=tl_tokens, posterior=False)
        navi_pred = model.navi_predictor(ag_valid=batch["sc/ag_valid"], ag_attr=batch["sc/ag_attr"], ag_motion=batch["sc/ag_motion"],
                                         ag_pose=batch["sc/ag_pose"], ag_type=batch["ref/ag_type"], **mp_tokens)
        return latent_post, latent_prior, navi_pred

    # ---- the reference's statements (:528-600)
    mp_tokens = model.mp_encoder(batch["sc/mp_valid"], batch["sc/mp_attr"], batch["sc/mp_pose"], batch["ref/mp_type"])
    tl_tokens = model.tl_encoder.pre_compute(tl_valid=batch["gt/tl_valid"], tl_attr=batch["sc/tl_attr"], tl_pose=batch["sc/tl_pose"], **mp_tokens)
    latent_post, latent_prior, navi_pred = encode_and_predict(mp_tokens, tl_tokens)
    ag_latent = None if latent_post is None else latent_post.sample(deterministic=True)
    ag_latent_valid = None if latent_post is None else latent_post.valid
    snap = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tl_tokens.items() if not k.startswith("_")}
    snap_mp = {k: v.clone() for k, v in mp_tokens.items() if torch.is_tensor(v)}
    buffer_reactive_replay = wm.reactive_replay(batch=batch, mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_latent=ag_latent,
                                                ag_latent_valid=ag_latent_valid, ag_navi=batch["gt/ag_navi"],
                                                ag_navi_valid=batch["gt/ag_valid"].any(-1), teacher_forcing=wm.teacher_forcing_reactive_replay,
                                                deterministic_action=True)
    torch.manual_seed(5)
    buffer_joint_future_pred = wm.joint_future_pred(batch=batch, mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_latent_dist=latent_prior,
                                                    ag_navi_dist=navi_pred, teacher_forcing=wm.teacher_forcing_joint_future_pred,
                                                    n_joint_future=wm.hparams.n_joint_future_wosac)
    eng = wm._engine
    assert eng.n == K and eng.tl_div == K, "the K rollouts of the scene share its lights and map tables"
    # the caller's dicts are as pre_compute / mp_encoder returned them (the reference overwrites tl_tokens in place, :460-462 - after
    # reactive_replay has consumed it; here neither call may change it)
    for k, v in snap.items():
        if torch.is_tensor(v):
            assert tl_tokens[k].shape == v.shape and torch.equal(tl_tokens[k], v), k
        else:
            assert tl_tokens[k] == v, k
    for k, v in snap_mp.items():
        assert torch.equal(mp_tokens[k], v), k
    A, T = sizes[0], wm.hparams.time_step_end
    assert buffer_reactive_replay.pred_pose.shape == (1, 1, A, T, 3)
    assert buffer_joint_future_pred.pred_pose.shape == (1, K, A, T, 3) and buffer_joint_future_pred.log_prob.shape == (1, K, A)
    assert torch.isfinite(buffer_joint_future_pred.pred_pose).all() and torch.isfinite(buffer_joint_future_pred.log_prob).all()
    free = slice(wm.hparams.time_step_current + 2, None)
    assert float((buffer_joint_future_pred.pred_pose[0, 0, :, free] - buffer_joint_future_pred.pred_pose[0, 1, :, free]).abs().max()) > 1e-3
    # ---- the same through this package's own helper (the path the oracle comparisons drive): bit-identical
    wm.engine_cache = 0  # (fresh engines: nothing of the first pass is reused)
    mp2, tl2 = wm.encode_scene(batch, tl_valid_key="gt/tl_valid")
    post2, _, _ = encode_and_predict(mp2, tl2)
    rr2 = wm.reactive_replay(batch, mp2, tl2, post2.sample(deterministic=True), post2.valid, batch["gt/ag_navi"], batch["gt/ag_valid"].any(-1),
                             wm.teacher_forcing_reactive_replay, True)
    _buffers_equal(buffer_reactive_replay, rr2, "reactive_replay")
    mp3, tl3 = wm.encode_scene(batch, tl_valid_key="gt/tl_valid", n_rollout=K)
    _, prior3, navi3 = encode_and_predict(mp2, tl2)
    torch.manual_seed(5)
    jf3 = wm.joint_future_pred(batch, mp3, tl3, prior3, navi3, wm.teacher_forcing_joint_future_pred, K)
    _buffers_equal(buffer_joint_future_pred, jf3, "joint_future_pred")
    assert torch.equal(buffer_joint_future_pred.log_prob, jf3.log_prob)


def test_forward_after_the_references_rollout_prologue(tb):
    """The body of the reference's `rollout` (waymo_motion.py:218-311) written out against this package with NO call the reference
    does not have: teacher_forcing.init -> self.dynamics.init(tl_state=tl_state_gt, **ag_tokens) -> self.model.init() -> per step
    teacher_forcing.get -> self.forward(mp_tokens, tl_tokens, ag_override, tl_override, player_override, deterministic_action) ->
    rule_checker.check (the reference's constructor arguments, ag_goal included; its sixteen result entries) ->
    self.diffbar_reward.get -> -pred_tl_state_dist.log_prob -> rollout_buffer.add -> dynamics.disable_ag / disable_navi. The first
    `forward` builds the device state from what Dynamics.init was given. Trajectories, validity, overrides and the feeding-back flags
    equal the engine's own rollout (`reactive_replay`) bit for bit; the reward and light NLL - recomputed here by the stand-alone
    entry points instead of read from tbx_sim_step's log - to float round-off."""
    dev = torch.device(DEV)
    wm, P, b, batch = _setup(tb, dev, (8, 64, 8), 4)
    RC = import_module("trafficbots_amd.utils.traffic_rule_checker")
    BUF = import_module("trafficbots_amd.utils.buffer")
    model = wm.model
    mp_tokens = model.mp_encoder(batch["sc/mp_valid"], batch["sc/mp_attr"], batch["sc/mp_pose"], batch["ref/mp_type"])
    tl_tokens = model.tl_encoder.pre_compute(tl_valid=batch["gt/tl_valid"], tl_attr=batch["sc/tl_attr"], tl_pose=batch["sc/tl_pose"], **mp_tokens)
    g = torch.Generator().manual_seed(2)
    ag_latent = torch.randn(1, 8, 16, generator=g).to(dev)
    valid = batch["gt/ag_valid"].any(-1)
    step_end = wm.hparams.time_step_end
    fast = wm.reactive_replay(batch, mp_tokens, tl_tokens, ag_latent, valid, batch["gt/ag_navi"], valid, wm.teacher_forcing_reactive_replay, True)
    # ---- reactive_replay's body (:399-436) ...
    rule_checker = RC.TrafficRuleChecker(mp_boundary=batch["map/boundary"], mp_valid=batch["map/valid"], mp_type=batch["map/type"],
                                         mp_pos=batch["map/pos"], mp_dir=batch["map/dir"], ag_type=batch["ref/ag_type"], ag_size=batch["ref/ag_size"],
                                         ag_goal=batch["agent/goal"], ag_dest=batch["agent/dest"], tl_valid=tl_tokens["tl_token_valid"],
                                         tl_pose=tl_tokens["tl_token_pose"], disable_check=wm.training)
    ag_tokens = {"ag_type": batch["ref/ag_type"], "ag_size": batch["ref/ag_size"], "ag_attr": batch["sc/ag_attr"], "gt_valid": batch["gt/ag_valid"],
                 "gt_pose": batch["gt/ag_pose"], "gt_motion": batch["gt/ag_motion"], "ag_latent": ag_latent, "ag_latent_valid": valid,
                 "ag_navi": batch["gt/ag_navi"], "ag_navi_valid": valid, "ag_navi_log_prob": torch.zeros_like(batch["sc/ag_attr"][:, :, 0])}
    tl_state_gt, teacher_forcing, deterministic_action = batch["gt/tl_state"], wm.teacher_forcing_reactive_replay, True
    # ---- ... and rollout's (:218-311), with `self` = wm
    teacher_forcing.init(ag_valid=ag_tokens["gt_valid"], ag_pose=ag_tokens["gt_pose"], ag_motion=ag_tokens["gt_motion"], tl_state=tl_state_gt,
                         current_epoch=wm.current_epoch)
    wm.dynamics.init(tl_state=tl_state_gt, **ag_tokens)
    wm.model.init()
    rollout_buffer = BUF.RolloutBuffer(step_end, wm.hparams.time_step_current)
    rollout_buffer.add_navi_log_prob(ag_tokens["ag_navi_log_prob"], ag_tokens["ag_navi_valid"])
    for _step in range(1, step_end + 1):
        ag_override, tl_override = teacher_forcing.get(_step, wm.dynamics.ag_valid, wm.dynamics.ag_pose, wm.dynamics.ag_motion)
        player_override = None
        pred_dict, vis_dict = wm.forward(mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_override=ag_override, tl_override=tl_override,
                                         player_override=player_override, deterministic_action=deterministic_action)
        violation = rule_checker.check(pred_dict["pred_valid"], pred_dict["pred_pose"], pred_dict["pred_motion"], wm.dynamics.tl_state)
        if _step == 1:
            assert set(violation) == {a + s for a in ("outside_map", "collided", "collided_wosac", "run_road_edge", "run_red_light", "passive",
                                                      "goal_reached", "dest_reached") for s in ("", "_this_step")}
        if _step >= ag_tokens["gt_valid"].shape[-1]:
            _gt_valid, _gt_pose, _gt_motion = None, None, None
        else:
            _gt_valid = ag_tokens["gt_valid"][:, :, _step]
            _gt_pose, _gt_motion = ag_tokens["gt_pose"][:, :, _step], ag_tokens["gt_motion"][:, :, _step]
        diffbar_reward = wm.diffbar_reward.get(pred_valid=pred_dict["pred_valid"], pred_pose=pred_dict["pred_pose"], pred_motion=pred_dict["pred_motion"],
                                               gt_valid=_gt_valid, gt_pose=_gt_pose, gt_motion=_gt_motion, ag_size=ag_tokens["ag_size"])
        if _step >= tl_state_gt.shape[2]:
            tl_state_nll = torch.zeros_like(tl_tokens["tl_token_pose"][:, :, 0])
            tl_state_nll_invalid = torch.ones_like(tl_tokens["tl_token_invalid"])
        else:
            _gt_tl_state = tl_state_gt[:, :, _step].max(-1)[1]
            tl_state_nll = -1.0 * (pred_dict["pred_tl_state_dist"].log_prob(_gt_tl_state))
            tl_state_nll_invalid = tl_tokens["tl_token_invalid"]
        rollout_buffer.add(violation=violation, diffbar_reward=diffbar_reward, tl_state_nll=tl_state_nll, tl_state_nll_invalid=tl_state_nll_invalid,
                           vis_dict=vis_dict, ag_override=ag_override, **pred_dict)
        wm.dynamics.disable_ag(violation, _gt_valid)
        wm.dynamics.disable_navi(violation)
    rollout_buffer.finish()
    rollout_buffer.flatten_joint_future(1)
    slow = rollout_buffer
    for name in ("pred_valid", "pred_pose", "pred_motion", "tl_state_nll_invalid", "action_log_prob", "mask_teacher_forcing", "navi_log_prob",
                 "navi_log_prob_valid"):
        assert torch.equal(getattr(slow, name), getattr(fast, name)), name
    for k in fast.violation:
        assert torch.equal(slow.violation[k], fast.violation[k]), k
    assert not slow.violation["goal_reached"].all() and slow.violation["goal_reached"].shape == fast.violation["outside_map"].shape
    assert torch.equal(slow.vis_dict["action"], fast.vis_dict["action"]) and torch.equal(slow.vis_dict["tl_state"], fast.vis_dict["tl_state"])
    assert torch.equal(slow.diffbar_reward["diffbar_reward_valid"], fast.diffbar_reward["diffbar_reward_valid"])
    for k in ("diffbar_reward", "r_imitation_pos", "r_imitation_rot", "r_imitation_spd"):
        torch.testing.assert_close(slow.diffbar_reward[k], fast.diffbar_reward[k], rtol=1e-5, atol=1e-6)
    inv = fast.tl_state_nll_invalid
    torch.testing.assert_close(slow.tl_state_nll.masked_fill(inv, 0), fast.tl_state_nll.masked_fill(inv, 0), rtol=1e-5, atol=1e-6)
    # the packaged loop (rollout(..., stepwise=True)) is this same sequence: bit-identical to the engine's log incl. reward and NLL
    packaged = wm.rollout(ag_tokens, mp_tokens, tl_tokens, tl_state_gt, teacher_forcing, wm._rule_checker(batch, batch["gt/ag_navi"], tl_tokens),
                          step_end, True, stepwise=True)
    packaged.flatten_joint_future(1)
    _buffers_equal(packaged, fast, "rollout(stepwise=True)")
    # a second rollout needs a second prologue; stepping past the log raises instead of writing nowhere
    with pytest.raises(RuntimeError):
        for _ in range(step_end + 1):
            wm.forward(mp_tokens, tl_tokens, ag_override, tl_override)


def test_wosac_shape_whole_horizon_damped_policy(tb):
    """BASELINE config 5 over its WHOLE horizon (the 32 x 128 shape had only been compared over 16 of its 90
    steps): 10 teacher-forced + 80 free-running steps of 32 rollouts x 128 agents through `joint_future_pred` in the reference's
    calling order, with the damped action head (x0.02: the random-weight loop is chaotic otherwise):
      * three rollouts point-wise against the oracle's Sim.rollout over all 90 steps (validity, feeding-back flags and light states
        identical; poses / motion / actions / light NLL bounded);
      * ALL 32 rollouts: light states identical to the oracle's (the lights' recurrence reads no agent), outside-map /
        destination-reached flags identical to the oracle's checks evaluated on the logged trajectories, step by step;
      * every fourth rollout: the five metric rule flags of all 90 steps bit-exact against the oracle's checks on the logged trajectories."""
    dev = torch.device(DEV)
    K, A = 32, 128
    ks = [0, 13, 31]
    wm, P, b, bd = _setup(tb, dev, (A, 1024, 128), 32)
    _damp(wm, P)
    T = wm.hparams.time_step_end
    D = import_module("trafficbots_amd.models.modules.distributions")
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=32), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    valid = bd["sc/ag_valid"].any(-1)
    onehot = torch.nn.functional.one_hot(bd["gt/ag_navi"], 1024).float()
    wm.hp.joint_future_pred_deterministic_k0 = False
    mp_tokens = wm.model.mp_encoder(bd["sc/mp_valid"], bd["sc/mp_attr"], bd["sc/mp_pose"], bd["ref/mp_type"])
    tl_tokens = wm.model.tl_encoder.pre_compute(tl_valid=bd["sc/tl_valid"], tl_attr=bd["sc/tl_attr"], tl_pose=bd["sc/tl_pose"], **mp_tokens)
    lat = D.DiagGaussian(torch.zeros(1, A, 16, device=dev), torch.zeros(16, device=dev), valid=valid)  # std-normal prior
    torch.manual_seed(21)
    buf = wm.joint_future_pred(batch=bd, mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_latent_dist=lat,
                               ag_navi_dist=D.DestCategorical(probs=onehot, valid=valid), teacher_forcing=wm.teacher_forcing_joint_future_pred,
                               n_joint_future=K)
    eng = wm._engine
    assert eng.tl_div == K and eng.n == K and buf.pred_pose.shape == (1, K, A, T, 3)
    z_all, dest_all = eng.ag_latent.view(K, A, 16).cpu(), eng.dest.cpu()
    # ---- three rollouts vs the oracle, all T steps (one oracle batch of three scene copies)
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    rep = lambda t, m: t.repeat_interleave(m, 0) if torch.is_tensor(t) and t.shape[0] == 1 else t
    vc = valid.cpu()
    with torch.no_grad():
        ro = O.Sim(om, scfg, False).rollout({k: rep(v, 3) for k, v in bh.items()}, {k: rep(v, 3) for k, v in mp_o.items()},
                                            {k: rep(v, 3) for k, v in tl_o.items()}, z_all[ks], vc.expand(3, -1), dest_all[ks], vc.expand(3, -1),
                                            scfg.teacher_forcing_joint_future_pred, T, gt_prefix="hist", tl_gt_key="sc/tl_state")
    dmax = lambda x, y: float((x.cpu() - y).abs().max())
    print(f"[wosac shape, damped, {T} steps, 3 rollouts vs oracle] max |d pose| {dmax(buf.pred_pose[0, ks], ro['pred_pose']):.3g}, |d motion| "
          f"{dmax(buf.pred_motion[0, ks], ro['pred_motion']):.3g}, |d action| {dmax(buf.vis_dict['action'][0, ks], ro['action']):.3g}, "
          f"|d nll| {dmax(buf.tl_state_nll[0, ks], ro['tl_state_nll']):.3g}")
    assert torch.equal(buf.pred_valid[0, ks].cpu(), ro["pred_valid"])
    assert torch.equal(buf.vis_dict["tl_state"][0, ks].cpu(), ro["tl_state"])
    assert torch.equal(buf.violation["outside_map"][0, ks].cpu(), ro["outside_map"])
    assert torch.equal(buf.violation["dest_reached"][0, ks].cpu(), ro["dest_reached"])
    # bounds <= 2 x measured on MI355X (0.0145 m-or-rad / 0.0114 / 0.0114 / 1.0e-4: two of 103,680 motion entries beyond 5e-3, both one
    # agent's yaw rate late in the horizon - a K-nearest set that flips on a near-tie moves an action by ~1e-2 even under the damped head)
    torch.testing.assert_close(buf.pred_pose[0, ks].cpu(), ro["pred_pose"], rtol=0, atol=3e-2)
    torch.testing.assert_close(buf.pred_motion[0, ks].cpu(), ro["pred_motion"], rtol=0, atol=2.5e-2)
    torch.testing.assert_close(buf.vis_dict["action"][0, ks].cpu(), ro["action"], rtol=0, atol=2.5e-2)
    torch.testing.assert_close(buf.tl_state_nll[0, ks].cpu(), ro["tl_state_nll"], rtol=1e-3, atol=2e-4)
    # ... and tight over the first 40 steps (10 warm-start + 30 free), where no such flip has happened yet
    h40 = slice(0, 40)
    torch.testing.assert_close(buf.pred_pose[0, ks][:, :, h40].cpu(), ro["pred_pose"][:, :, h40], rtol=1e-4, atol=5e-3)
    torch.testing.assert_close(buf.vis_dict["action"][0, ks][:, :, h40].cpu(), ro["action"][:, :, h40], rtol=1e-3, atol=5e-3)
    free = slice(wm.hparams.time_step_current + 10, None)
    assert dmax(buf.pred_pose[0, 0, :, free], buf.pred_pose[0, 13, :, free].cpu()) > 1e-3  # (the rollouts differ: their latents / destinations do)
    # ---- all 32 rollouts: lights and the feeding-back flags
    assert torch.isfinite(buf.pred_pose).all() and torch.isfinite(buf.vis_dict["action"]).all()
    tl_all = buf.vis_dict["tl_state"][0].cpu()  # [K, L, T, 5]
    assert torch.equal(tl_all, ro["tl_state"][:1].expand(K, -1, -1, -1)), "the lights' recurrence reads no agent: one trajectory for all rollouts"
    dest = O.Sim.dest_info(dest_all, rep(b["map/valid"], K), rep(b["map/type"], K), rep(b["map/pos"], K), rep(b["map/dir"], K))
    bnd = rep(b["map/boundary"], K)
    pv, pp = buf.pred_valid[0].cpu(), buf.pred_pose[0].cpu()  # [K, A, T(, 3)]
    outside, reached = torch.zeros(K, A, dtype=torch.bool), torch.zeros(K, A, dtype=torch.bool)
    for t in range(T):
        out_now, reach_now = O.Sim.feedback_checks(pv[:, :, t], pp[:, :, t], bnd, dest, reached)
        outside, reached = outside | out_now, reached | reach_now
        assert torch.equal(buf.violation["outside_map"][0, :, :, t].cpu(), outside), t
        assert torch.equal(buf.violation["dest_reached"][0, :, :, t].cpu(), reached), t
    # ---- every fourth rollout: the five metric flags, all steps
    sub = list(range(0, K, 4))
    r = lambda t: t.repeat_interleave(len(sub), 0).cpu()
    o = R.RuleCheckOracle(r(b["map/valid"]), r(b["map/type"]), r(b["map/pos"]), r(b["map/dir"]), r(b["ref/ag_type"]), r(b["ref/ag_size"]),
                          r(tl_tokens["tl_token_valid"]), r(tl_tokens["tl_token_pose"]))
    pick = lambda t: t[0, sub].cpu()
    pvs, pps, pms, tss = pick(buf.pred_valid), pick(buf.pred_pose), pick(buf.pred_motion), pick(buf.vis_dict["tl_state"])
    n_flag = 0
    for t in range(T):
        v = o.check(pvs[:, :, t], pps[:, :, t], pms[:, :, t], tss[:, :, t])
        for key, x in v.items():
            assert torch.equal(pick(buf.violation[key])[:, :, t], x), (key, t)
            n_flag += int(x.sum()) if key.endswith("_this_step") else 0
    assert n_flag > 0  # (the comparison is not of all-false flags)
