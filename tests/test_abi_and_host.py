"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/tbx_hip.h
declares, argument validation returns error codes (no compute without a GPU), state-dict layout equals the reference's,
host-side schedules (teacher forcing masks, scene-centric re-keying) equal the oracle's."""
import ctypes as C
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import trafficbots_oracle as O


@pytest.fixture(scope="module")
def hip(tb):
    h = import_module("trafficbots_amd.hip")
    h.load()
    return h


def test_library_exports_every_declared_symbol(hip):
    lib = hip.load()
    syms = hip.declared_symbols()
    assert set(syms) >= {"tbx_version", "tbx_error_string", "tbx_knn_embed", "tbx_pose_embed", "tbx_knarpe_attn_fwd",
                         "tbx_rowchain", "tbx_rowchain_ex", "tbx_knarpe_attn_bwd", "tbx_agent_prep", "tbx_tl_prep", "tbx_map_prep", "tbx_sim_step"}
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.tbx_version() == 4
    assert lib.tbx_error_string(-2).decode().startswith("shape")


def test_argument_validation_returns_codes_without_a_gpu(hip):
    lib = hip.load()
    # null pointers / bad sizes are rejected before any launch
    assert lib.tbx_knn_embed(None, None, None, None, 1, 1, 1, 1, 1, 1.0, None, None, None, None, None, None, 128, None) == -1
    assert lib.tbx_rowchain(None, 0, 0, 0, 16, 132, None) == -1
    st = (hip.Stage * 1)(hip.Stage(op=hip.OP_LINEAR, src=0, dst=1, k=128, n=128, ld=128))
    assert lib.tbx_rowchain(st, 1, 16, 0, 24, 132, None) == -2      # tile_rows must be 16, 32 or 48
    assert lib.tbx_rowchain(st, 1, 16, 0, 16, 130, None) == -3      # ldw % 4
    assert lib.tbx_rowchain(st, 1, 16, 0, 16, 132, None) == -1      # LINEAR without a weight pointer
    assert lib.tbx_sim_step(None, None) == -1
    with pytest.raises(RuntimeError):
        hip.pose_embed(torch.zeros(4, 3), torch.zeros(32), torch.zeros(64), 128)  # CPU tensor: no fallback path


def test_state_dict_layout_equals_reference(tb, golden_dir):
    M = import_module("trafficbots_amd.models.traffic_bots")
    model = M.TrafficBots(**tb.config.default_model_cfg())
    want = dict(l.split(" ", 1) for l in (golden_dir / "state_dict_keys.txt").read_text().strip().split("\n"))
    got = {k: str(tuple(v.shape)) for k, v in model.state_dict().items()}
    assert got == want
    assert sum(p.numel() for p in model.parameters()) == 10657094
    # a reference-shaped (Lightning-prefixed) checkpoint loads strictly
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    ckpt = {"model." + k: v for k, v in model.state_dict().items()}
    missing, unexpected = wm.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    opt, sch = wm.configure_optimizers()
    assert len(opt[0].param_groups) == 2 and opt[0].param_groups[0]["lr"] == 2e-4


def test_host_schedules_equal_oracle(tb):
    batch = tb.synthetic.make_scene(2, 8, 64, 8, seed=3)
    full = {**batch, **tb.synthetic.to_history_batch(batch)}
    SC = import_module("trafficbots_amd.data_modules.scene_centric").SceneCentricPreProcessing
    for training in (True, False):
        pp = SC(time_step_current=10, tl_mode="lane", navi_mode="dest", dropout_p_history=-1, data_size=tb.synthetic.DATA_SIZE)
        pp.train(training)
        b = pp({k: v.clone() for k, v in full.items()})
        bo = O.scene_centric(full, training=training)
        for k in bo:
            if k.startswith(("sc/", "gt/", "ref/")):
                assert torch.equal(b[k], bo[k]), k
    TF = import_module("trafficbots_amd.utils.teacher_forcing").TeacherForcing
    for cfg in (dict(step_spawn_agent=10, step_warm_start=10), dict(step_spawn_agent=90, step_warm_start=10),
                dict(step_spawn_agent=0, step_warm_start=-1)):
        tf = TF(**cfg)
        tf.init(bo["gt/ag_valid"], bo["gt/ag_pose"], bo["gt/ag_motion"], bo["gt/tl_state"], 0)
        assert torch.equal(tf.ag_teacher_forcing, O.Sim.teacher_forcing_mask(bo["gt/ag_valid"], **cfg))
        ag, tl = tf.get(5, None, None, None)
        assert torch.equal(ag["valid"], tf.ag_teacher_forcing[:, :, 5]) and bool(tl["valid"].all())
        ag, tl = tf.get(200, None, None, None)
        assert not bool(ag["valid"].any()) and not bool(tl["valid"].any())


def test_chain_program_encoding(hip):
    """The Chain builder encodes strides / groups / flags the way include/tbx_hip.h documents (host logic only)."""
    ch = hip.Chain(16, 132)
    w = torch.zeros(256, 128)
    ch._add(op=hip.OP_COPY, src=0, dst=1, n=4)
    st = ch.stages[0]
    assert (st.op, st.src, st.dst, st.n) == (hip.OP_COPY, 0, 1, 4)
    assert C.sizeof(hip.Stage) == 88 and C.sizeof(hip.AttnSeg) == 72


def test_ctypes_mirrors_have_the_layout_gcc_gives_the_header(hip, tmp_path):
    """sizeof of every struct that crosses the C ABI, as gcc lays out include/tbx_hip.h, equals the ctypes mirror's."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    pairs = [("tbx_stage_t", hip.Stage), ("tbx_attn_seg_t", hip.AttnSeg), ("tbx_dec_mid_t", hip.DecMid), ("tbx_dec_layer_t", hip.DecLayer), ("tbx_heads_tail_t", hip.HeadsTail), ("tbx_knn_job_t", hip.KnnJob), ("tbx_pose_embed_job_t", hip.PoseEmbedJob), ("tbx_sim_state_t", hip.SimState),
             ("tbx_train_chain_t", hip.TrainChainArgs), ("tbx_rule_ctx_t", hip.RuleCtx), ("tbx_layer_tile_t", hip.LayerTile),
             ("tbx_heads_tile_t", hip.HeadsTile), ("tbx_window_tile_t", hip.WindowTile), ("tbx_agent_prep_args_t", hip.AgentPrepArgs), ("tbx_front_t", hip.Front), ("tbx_tl_tail_t", hip.TlTail), ("tbx_pack_job_t", hip.PackJob)]
    src = tmp_path / "sz.c"
    # ... and offsetof of every field (same names on both sides): runs of same-sized pointers keep sizeof when two fields swap
    fields = [(c, t, f[0]) for c, t in pairs for f in t._fields_]
    src.write_text('#include "tbx_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({c}));' for c, _ in pairs) +
                   "".join(f'printf("%zu\\n", offsetof({c}, {f}));' for c, _, f in fields) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(root / "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[:len(pairs)] == [C.sizeof(t) for _, t in pairs]
    for (c, t, f), off in zip(fields, out[len(pairs):]):
        assert getattr(t, f).offset == off, (c, f, getattr(t, f).offset, off)


def test_new_entry_points_validate_arguments_without_a_gpu(hip):
    lib = hip.load()
    assert lib.tbx_keyed_dropout(None, None, 4, 4, 2, 0.1, None, 0, 1, 0, None) == -1
    assert lib.tbx_linear_wgrad_splits(0, 128, 128) == -1
    assert 1 <= lib.tbx_linear_wgrad_splits(2_000_000, 128, 128) <= 1024
    assert lib.tbx_linear_wgrad_splits(100, 128, 128) == 2  # at least 64 rows per split
    assert lib.tbx_linear_wgrad(None, 128, None, 128, 1000, 128, 128, None, None, None, 4, None) == -1
    assert lib.tbx_train_chain_fwd(None, None, 0, 0, 0, 1, None) == -1
    args = hip.TrainChainArgs()
    args.n_batch, args.n_ag, args.n_step, args.n_step_gt, args.n_node, args.window = 1, 4, 10, 11, 20, 11
    assert lib.tbx_train_chain_fwd(C.byref(args), None, 8, 0, 0, 1, None) == -1  # null state pointers
    assert lib.tbx_train_chain_bwd(C.byref(args), None, 8, 8, None, None, None) == -1


def test_group_tile_rows_picks_the_least_wasteful_tile(hip):
    assert hip.group_tile_rows(11, 64) == 16          # small grid: one window per 16-row tile (shortest critical path)
    assert hip.group_tile_rows(11, 4096) == 48        # 4 windows in 48 rows (92 %) beat 2 in 32 (69 %)
    assert hip.group_tile_rows(20, 4096) == 48        # 2 polylines in 48 rows beat 1 in 32
    assert hip.group_tile_rows(20, 100) == 32
    assert hip.group_tile_rows(16, 4096) == 32        # 2 x 16 fills 32 rows exactly
    assert hip.group_tile_rows(5, 4096) == 32         # 6 in 32 (94 %) vs 9 in 48 (94 %): the smaller tile


def test_lights_per_scene_view_of_expanded_light_tokens(tb):
    """RolloutEngine steps the lights once per scene: the per-scene view of tokens that encode_scene expanded per rollout."""
    E = import_module("trafficbots_amd.utils.rollout_engine")
    n_scene, K, L, M = 3, 4, 5, 7
    g = torch.Generator().manual_seed(0)
    per_scene = {"tl_token_pose": torch.randn(n_scene, L, 3, generator=g), "tl_token_valid": torch.rand(n_scene, L, generator=g) > 0.3,
                 "knn_idx_tl2mp": torch.randint(0, M, (n_scene, L, 2), generator=g), "tl_token_attr": torch.randn(n_scene, L, 8, generator=g)}
    expanded = {k: v.repeat_interleave(K, 0) for k, v in per_scene.items()}
    expanded.update(mp_batch_div=K, n_mp=M, mp_feat_flat=torch.randn(n_scene * M, 8, generator=g), _kv_mp={"cached": 1})
    view = E.lights_per_scene(expanded, K)
    for k, v in per_scene.items():
        assert torch.equal(view[k], v), k
    assert view["mp_batch_div"] == 1 and view["tl_batch_div"] == K and view["ag_mp_batch_div"] == K and view["n_mp"] == M
    assert view["mp_feat_flat"] is expanded["mp_feat_flat"] and "_kv_mp" not in view


def test_tl_nll_all_steps_equals_the_per_step_loop(tb):
    """train_graph.tl_nll_all_steps (waymo_motion.py:270-283 for every step at once) vs the step loop with Categorical."""
    TG = import_module("trafficbots_amd.train_graph")
    from torch.distributions import Categorical

    g = torch.Generator().manual_seed(1)
    n, T, L, Tt = 2, 12, 5, 9  # ground truth ends before the rollout does
    logits = torch.randn(n, T, L, 5, generator=g)
    tl_gt = torch.nn.functional.one_hot(torch.randint(0, 5, (n, L, Tt), generator=g), 5).bool()
    inv = torch.rand(n, L, generator=g) < 0.3
    nll, nll_inv = TG.tl_nll_all_steps(logits, tl_gt, inv)
    for step in range(1, T + 1):
        if step < Tt:
            want = -Categorical(logits=logits[:, step - 1]).log_prob(tl_gt[:, :, step].max(-1)[1])
            torch.testing.assert_close(nll[:, :, step - 1], want)
            assert torch.equal(nll_inv[:, :, step - 1], inv)
        else:
            assert float(nll[:, :, step - 1].abs().max()) == 0.0 and bool(nll_inv[:, :, step - 1].all())


def test_graphed_train_step_refuses_without_the_ordered_memset_path(tb, monkeypatch):
    """hipGraph memset nodes replay out of order on ROCm's AQL-packet fast path (stale bias gradients): GraphedTrainStep must
    refuse to capture unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the environment (pl_modules/data_parallel.py)."""
    from importlib import import_module

    import pytest

    DP = import_module("trafficbots_amd.pl_modules.data_parallel")
    for bad in (None, "1", ""):
        if bad is None:
            monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
        else:
            monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", bad)
        with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"):
            DP.GraphedTrainStep(object(), object(), {})


def test_schedule_switches_follow_the_environment_and_replace(tb, monkeypatch):
    """engine.Schedule: every boolean switch is on by default except the opt-ins (measured slower: front_big, attn_fold_big, pool_proj;
    a different arithmetic: kv_bf16, split_bf16, attn_mfma, linear_bf16), `TBX_<NAME>=0 / 1` flips it in from_env(), replace() leaves the original alone, and a
    schedule is a value (two equal ones compare equal: engines are cached by it)."""
    import dataclasses
    from importlib import import_module

    E = import_module("trafficbots_amd.engine")
    for k in list(__import__("os").environ):
        if k.startswith("TBX_"):
            monkeypatch.delenv(k)
    d = E.Schedule.from_env()
    off = {"front_big", "attn_fold_big", "pool_proj", "kv_bf16", "split_bf16", "attn_mfma", "linear_bf16"}
    bools = [f.name for f in dataclasses.fields(E.Schedule) if isinstance(getattr(d, f.name), bool)]
    assert {"knn_aux_big", "prime_graph", "fused_tail", "front_fused", "dec_tail_mfma", "front_big"} <= set(bools)
    for name in bools:
        assert getattr(d, name) == (name not in off), name
    assert d == E.Schedule.from_env() and hash(dataclasses.astuple(d)) == hash(dataclasses.astuple(E.Schedule.from_env()))
    for name, env in (("knn_aux_big", "TBX_KNN_AUX_BIG"), ("prime_graph", "TBX_PRIME_GRAPH"), ("fused_tail", "TBX_FUSED_TAIL")):
        monkeypatch.setenv(env, "0")
        assert getattr(E.Schedule.from_env(), name) is False
        monkeypatch.delenv(env)
    monkeypatch.setenv("TBX_FRONT_BIG", "1")
    assert E.Schedule.from_env().front_big is True
    r = d.replace(knn_aux_big=False)
    assert r.knn_aux_big is False and d.knn_aux_big is True and r != d


def test_training_metrics_class_equals_the_captured_steps_loss(tb):
    """models/metrics/training.py::TrainingMetrics (the reference-shaped accumulator: update(buffer, ...) / compute()) and
    train_graph.training_loss (the fused expression the captured training step uses, itself checked against the reference's golden
    loss dict in tests/test_hip_training.py) give the same numbers on the same rollout log - both switch settings of
    loss_for_teacher_forcing, and a batch without a valid light drops its term the way the reference's `if counter > 0` does."""
    from importlib import import_module
    from types import SimpleNamespace

    TG = import_module("trafficbots_amd.train_graph")
    TM = import_module("trafficbots_amd.models.metrics.training")
    D = import_module("trafficbots_amd.models.modules.distributions")
    g = torch.Generator().manual_seed(0)
    n, A, T, L, M = 3, 9, 30, 5, 20
    r = lambda *s: torch.rand(*s, generator=g)
    for tf_loss, no_lights in ((True, False), (False, False), (True, True)):
        cfg = tb.config.default_sim_cfg()["training_metrics"]
        cfg["loss_for_teacher_forcing"] = tf_loss
        ro = dict(pred_valid=r(n, A, T) < 0.8, tf=r(n, A, T) < 0.2, reward_valid=r(n, A, T) < 0.9, reward=-r(n, A, T),
                  tl_nll_invalid=(r(n, L, T) < 0.3) | no_lights, tl_nll=r(n, L, T))
        post = D.DiagGaussian(torch.randn(n, A, 16, generator=g), torch.randn(n, A, 16, generator=g) * 0.3, valid=r(n, A) < 0.9)
        prior = D.DiagGaussian(torch.zeros(n, A, 16), torch.zeros(16), valid=r(n, A) < 0.8)
        navi = D.DestCategorical(logits=torch.randn(n, A, M, generator=g), valid=r(n, A) < 0.9)
        navi_gt = torch.randint(0, M, (n, A), generator=g)
        want = TG.training_loss(cfg, ro, navi, navi_gt, post, prior)
        buf = SimpleNamespace(pred_valid=ro["pred_valid"], mask_teacher_forcing=ro["tf"], tl_state_nll=ro["tl_nll"],
                              tl_state_nll_invalid=ro["tl_nll_invalid"],
                              diffbar_reward={"diffbar_reward_valid": ro["reward_valid"], "diffbar_reward": ro["reward"]})
        m = TM.TrainingMetrics(prefix="training", train_navi=True, train_latent=True, **cfg)
        got = m(buf, None, navi, navi_gt, post, prior)
        for k in ("loss", "vae_kl", "diffbar_reward", "navi_loss"):
            torch.testing.assert_close(got[f"training/{k}"], want[k], rtol=1e-6, atol=1e-6)
        if no_lights:
            assert "training/tl_state_loss" not in got and float(want["tl_state_loss"]) == 0.0
        else:
            torch.testing.assert_close(got["training/tl_state_loss"], want["tl_state_loss"], rtol=1e-6, atol=1e-6)


def test_multi_tensor_copy_of_the_engine_refill(tb):
    """RolloutEngine._copy_all (the ~80 copies of a scene commit as a few multi-tensor launches): same-dtype contiguous pairs of equal
    shape are grouped by dtype, everything else - dtype conversions, strided views, broadcasts - falls back to copy_, aliased pairs
    are skipped; every destination ends equal to its source."""
    R = import_module("trafficbots_amd.utils.rollout_engine").RolloutEngine
    g = torch.Generator().manual_seed(0)
    f = lambda *s: torch.randn(*s, generator=g)
    same = f(4, 3)
    base = torch.zeros(6, 8)
    pairs = [(torch.zeros(4, 3), f(4, 3)), (torch.zeros(5), f(5)),                                  # float32, contiguous: one group
             (torch.zeros(7, dtype=torch.uint8), (torch.rand(7, generator=g) < 0.5).to(torch.uint8)),  # uint8: another group
             (torch.zeros(2, 3, dtype=torch.int64), torch.randint(0, 9, (2, 3), generator=g)),          # int64
             (torch.zeros(4, dtype=torch.float32), torch.randint(0, 9, (4,), generator=g)),            # conversion: copy_
             (base[:, ::2], f(6, 4)),                                                                 # strided destination: copy_
             (torch.zeros(3, 4), f(1, 4)),                                                            # broadcast source: copy_
             (same, same)]                                                                            # aliased: skipped
    want = [s.clone() for _, s in pairs]
    R._copy_all(pairs)
    for (d, _), w in zip(pairs, want):
        assert torch.equal(d, w.to(d.dtype).expand_as(d))
    R._copy_all([])  # (nothing to do)


def test_training_flags_the_time_batched_step_cannot_honour_raise(tb):
    """training_detach_model_input / training_deterministic_action (waymo_motion.py:158-161, :370): the training step batches the
    decoder over time, exact only with detached inputs and deterministic actions (the defaults). False used to be accepted and
    ignored - plausible, wrong gradients; now the constructor refuses, like every other non-default branch."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    for flag in ("training_detach_model_input", "training_deterministic_action"):
        scfg = tb.config.default_sim_cfg()
        assert scfg[flag] is True
        scfg[flag] = False
        with pytest.raises(NotImplementedError, match=flag):
            W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **scfg)


def test_dynamics_init_holds_step_zero_until_the_first_forward(tb):
    """The reference's prologue `self.dynamics.init(tl_state=tl_state_gt, **ag_tokens)` (waymo_motion.py:228): between it and the
    first `forward` the state attributes read step 0 of the ground truth (teacher_forcing.get is handed them, :233-235)."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    g = torch.Generator().manual_seed(0)
    n, A, T, L = 2, 5, 11, 3
    tok = dict(gt_valid=torch.rand(n, A, T, generator=g) > 0.2, gt_pose=torch.randn(n, A, T, 3, generator=g), gt_motion=torch.randn(n, A, T, 3, generator=g),
               ag_type=torch.eye(3)[torch.randint(0, 3, (n, A), generator=g)].bool(), ag_attr=torch.randn(n, A, 6, generator=g),
               ag_size=torch.rand(n, A, 3, generator=g), ag_latent=torch.randn(n, A, 16, generator=g), ag_latent_valid=torch.ones(n, A, dtype=torch.bool),
               ag_navi=torch.randint(0, 7, (n, A), generator=g), ag_navi_valid=torch.rand(n, A, generator=g) > 0.5,
               ag_navi_log_prob=torch.zeros(n, A))
    tl = torch.eye(5)[torch.randint(0, 5, (n, L, T), generator=g)].bool()
    dyn = wm.dynamics
    with pytest.raises(RuntimeError):
        dyn.ag_pose
    dyn.init(tl_state=tl, **tok)
    wm.model.init()
    assert torch.equal(dyn.ag_valid, tok["gt_valid"][:, :, 0]) and torch.equal(dyn.ag_pose, tok["gt_pose"][:, :, 0])
    assert torch.equal(dyn.ag_motion, tok["gt_motion"][:, :, 0]) and torch.equal(dyn.tl_state, tl[:, :, 0])
    assert torch.equal(dyn.ag_navi_valid, tok["ag_navi_valid"]) and not dyn.mask_navi_reached.any() and not dyn.ag_disabled.any()
    assert dyn.ag_navi is tok["ag_navi"] and dyn.ag_type is tok["ag_type"]
    with pytest.raises(RuntimeError):  # no engine yet: the reference's order is init -> forward -> disable_*
        dyn.disable_navi({"dest_reached_this_step": torch.zeros(n, A, dtype=torch.bool)})
    tf = wm.teacher_forcing_reactive_replay
    tf.init(ag_valid=tok["gt_valid"], ag_pose=tok["gt_pose"], ag_motion=tok["gt_motion"], tl_state=tl, current_epoch=0)
    ag_override, tl_override = tf.get(1, dyn.ag_valid, dyn.ag_pose, dyn.ag_motion)
    assert ag_override["pose"].shape == (n, A, 3) and tl_override["state"].shape == (n, L, 5)


def test_light_tokens_per_rollout_is_a_copy_with_encode_scenes_layout(tb):
    """joint_future_pred's own expansion of pre_compute's light tokens (the reference's :458-462): every per-light tensor repeated K
    times along the batch, map targets still indexed per scene (mp_batch_div = K), caches dropped, the caller's dict untouched; the
    rollout engine's per-scene view (lights_per_scene) of the result is the original."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    E = import_module("trafficbots_amd.utils.rollout_engine")
    n_scene, K, L, M = 2, 4, 5, 7
    g = torch.Generator().manual_seed(1)
    tl = {"tl_token_pose": torch.randn(n_scene, L, 3, generator=g), "tl_token_valid": torch.rand(n_scene, L, generator=g) > 0.3,
          "knn_idx_tl2mp": torch.randint(0, M, (n_scene, L, 2), generator=g), "rel_tl2tl": torch.randn(n_scene, L, 2, 3, generator=g),
          "rpe_tl2tl": None, "mp_batch_div": 1, "n_mp": M, "mp_feat_flat": torch.randn(n_scene * M, 8, generator=g), "_kv_mp": {"cached": 1}}
    before = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tl.items()}
    ex = W.WaymoMotion._tl_tokens_per_rollout(tl, K)
    assert ex is not tl and "_kv_mp" not in ex and ex["mp_batch_div"] == K and ex["n_mp"] == M and ex["rpe_tl2tl"] is None
    assert ex["mp_feat_flat"] is tl["mp_feat_flat"]
    for k in ("tl_token_pose", "tl_token_valid", "knn_idx_tl2mp", "rel_tl2tl"):
        assert torch.equal(ex[k], tl[k].repeat_interleave(K, 0)), k
    for k, v in before.items():  # the caller's dict: same keys, same values
        assert (torch.equal(tl[k], v) if torch.is_tensor(v) else tl[k] == v), k
    view = E.lights_per_scene(ex, K)
    for k in ("tl_token_pose", "tl_token_valid", "knn_idx_tl2mp", "rel_tl2tl"):
        assert torch.equal(view[k], tl[k]), k


def test_rule_navi_check_validates_arguments_without_a_gpu(hip):
    lib = hip.load()
    assert lib.tbx_rule_navi_check(None, None, None, 1, None, None, None, None, None, None, None, 1, 1, 0, None, None, None) == -1
    one = C.c_void_p(16)  # (never dereferenced: validation happens before the launch)
    assert lib.tbx_rule_navi_check(one, one, one, 2, None, None, None, None, None, None, None, 3, 4, 0, one, one, None) == -1  # n % div
    assert lib.tbx_rule_navi_check(one, one, one, 1, one, None, None, None, None, None, None, 1, 4, 20, one, one, None) == -1  # partial dest tables
    assert lib.tbx_rule_navi_check(one, one, one, 1, None, None, None, None, None, one, None, 1, 4, 0, one, one, None) == -1   # goal without its threshold


def test_host_descriptor_paths_read_their_host_arrays_in_bounds(hip):
    """The entry points that take HOST arrays / structures (job lists, stage programs, nested descriptors) walk them on the host before
    any launch: driven here with well-formed host data and placeholder device addresses on a machine WITHOUT a GPU, each returns an
    error code (bad argument, or the launch error once validation has passed) - never a crash. Under tools/sanitize_host.sh
    (AddressSanitizer + UBSan on the host halves) these are the reads that get checked."""
    if torch.cuda.is_available():
        pytest.skip("placeholder device addresses: the no-GPU form of this test (a launch would dereference them)")
    lib = hip.load()
    dp = lambda i: 0x10000 * (i + 1)  # 16-byte aligned, distinct, never dereferenced on the host
    jobs = (hip.KnnJob * 3)()
    for j, (n_src, n_tgt, k) in enumerate(((64, 64, 25), (64, 1024, 64), (64, 128, 24))):
        jb = jobs[j]
        jb.src_pose, jb.src_invalid, jb.tgt_pose, jb.tgt_invalid = dp(8 * j), dp(8 * j + 1), dp(8 * j + 2), dp(8 * j + 3)
        jb.idx, jb.invalid, jb.rel_pose, jb.emb = dp(8 * j + 4), dp(8 * j + 5), dp(8 * j + 6), None
        jb.n_batch, jb.n_src, jb.n_tgt, jb.tgt_batch_div, jb.k, jb.dist_limit = 1, n_src, n_tgt, 1, k, 1500.0
    for n in (1, 2, 3):
        assert lib.tbx_knn_embed_multi(jobs, n, dp(40), dp(41), 128, None) < 0
    assert lib.tbx_knn_embed_multi(jobs, 0, dp(40), dp(41), 128, None) == -1
    jobs[1].k = 4096  # more neighbours than targets
    assert lib.tbx_knn_embed_multi(jobs, 3, dp(40), dp(41), 128, None) < 0
    pe = hip.PoseEmbedJob()
    pe.pose3, pe.freqs_xy, pe.freqs_yaw, pe.out, pe.n, pe.pe_dim, pe.ld_out, pe.col_off = dp(50), dp(51), dp(52), dp(53), 64, 128, 128, 0
    jobs[1].k = 64
    assert lib.tbx_knn_embed_multi_pe(jobs, 3, dp(40), dp(41), 128, C.byref(pe), None) < 0
    # a full stage program (MAX_STAGES entries) and one past it
    st = (hip.Stage * (hip.MAX_STAGES + 1))()
    for s in st:
        s.op, s.src, s.dst, s.k, s.n, s.ld, s.p0 = hip.OP_LINEAR, 0, 1, 128, 128, 128, dp(60)
    assert lib.tbx_rowchain(st, hip.MAX_STAGES, 16, 0, 16, 132, None) < 0
    assert lib.tbx_rowchain(st, hip.MAX_STAGES + 1, 16, 0, 16, 132, None) < 0
    # nested descriptors: zero-initialised (every pointer NULL) and partially filled
    for cls, fn in ((hip.DecLayer, lib.tbx_knarpe_dec_layer), (hip.LayerTile, lib.tbx_layer_tile), (hip.HeadsTile, lib.tbx_heads_tile),
                    (hip.WindowTile, lib.tbx_window_tile), (hip.Front, lib.tbx_front)):
        d = cls()
        assert fn(C.byref(d), None) < 0, cls.__name__
    fr = hip.Front()
    fr.jobs, fr.n_jobs = C.cast(jobs, C.c_void_p), 3
    assert lib.tbx_front(C.byref(fr), None) < 0
    # round 6: the paired launches (two nested descriptors each), the job array of the multi-image pack, the new glue entry points
    da, db = hip.DecLayer(), hip.DecLayer()
    assert lib.tbx_knarpe_dec_layer_pair(C.byref(da), C.byref(db), None) < 0 and lib.tbx_knarpe_dec_layer_pair(None, C.byref(db), None) == -1
    fa, fb = hip.Front(), hip.Front()
    assert lib.tbx_front_pair(C.byref(fa), C.byref(fb), None) < 0 and lib.tbx_front_pair(C.byref(fa), None, None) == -1
    pj = (hip.PackJob * 50)()  # (more than one launch's worth: 48 jobs per launch)
    for i, j in enumerate(pj):
        j.w, j.bias, j.out, j.n, j.k, j.ld, j.groups, j.wt = dp(70 + 2 * i), None, dp(71 + 2 * i), 128, 128, 128, 1, i & 1
    assert lib.tbx_pack_weight_mfma32_multi(pj, 0, None) == 0  # nothing to do
    assert lib.tbx_pack_weight_mfma32_multi(None, 3, None) == -1
    pj[49].k = 100  # an unsupported width in the LAST job: found by the host-side walk of its group
    assert lib.tbx_pack_weight_mfma32_multi(pj, 50, None) < 0
    pj[49].k, pj[7].out = 128, None
    assert lib.tbx_pack_weight_mfma32_multi(pj, 50, None) == -1
    assert lib.tbx_pair_bias_relu(None, dp(1), dp(2), 1, 4, 8, 128, 1, None) == -1 and lib.tbx_pair_bias_relu(dp(0), dp(1), dp(2), 1, 4, 8, 126, 1, None) == -1
    assert lib.tbx_pair_bias_relu(dp(0), dp(1), dp(2), 0, 4, 8, 128, 1, None) == 0  # an empty batch
    assert lib.tbx_layernorm_bwd_add(dp(0), dp(1), dp(2), dp(3), dp(4), 16, 64, dp(5), dp(6), dp(7), dp(8), dp(9), None) < 0  # cols != 128
    assert lib.tbx_layernorm_bwd_add(dp(0), dp(1), dp(2), dp(3), dp(4), 16, 128, 0x10004, dp(6), dp(7), dp(8), dp(9), None) < 0  # misaligned `add`
    ss = hip.SimState()
    assert lib.tbx_sim_step(C.byref(ss), None) == -1 and lib.tbx_sim_step_parts(C.byref(ss), 3, None) == -1
    rc = hip.RuleCtx()
    assert lib.tbx_rule_check(C.byref(rc), dp(1), dp(2), dp(3), dp(4), 1, 0, 1, dp(5), None) == -1


def test_clip_on_the_flat_gradient_buffer_equals_clip_grad_norm(tb):
    """pl_modules/data_parallel.clip_gradients on a FlatGrads (one 2-norm + one scale of the flat buffer) against
    torch.nn.utils.clip_grad_norm_ on the same gradients as a parameter list: same total norm, same clipped gradients (fp32 summation
    order apart), both when the norm exceeds the bound and when it does not."""
    DP = import_module("trafficbots_amd.pl_modules.data_parallel")
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 128), (128,), (640, 128), (5,), (1, 128)]
    for scale, max_norm in ((10.0, 5.0), (1e-3, 5.0)):
        pa = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        pb = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        grads = [torch.randn(s, generator=g) * scale for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        flat = DP.FlatGrads(pa)
        total = DP.clip_gradients(flat, max_norm)
        ref = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        torch.testing.assert_close(total, ref, rtol=1e-6, atol=0)
        for i, (p, q) in enumerate(zip(pa, pb)):
            assert p.grad.data_ptr() == flat.views[i].data_ptr()  # still the flat buffer's slice
            torch.testing.assert_close(p.grad, q.grad, rtol=2e-6, atol=0)
        assert DP.clip_gradients(flat, 0) is None
