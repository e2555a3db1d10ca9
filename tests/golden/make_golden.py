#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE (read-only at /root/reference).

Container-only tooling: the reference never travels to the GPU box; what is committed is data (outputs of the
reference on seeded inputs), plus this script. Inputs are regenerated from seeds by `synthetic.make_scene`, weights
by `det_fill` (both ours, shipped), so fixtures hold only the reference's outputs.

Shims (no source edits): omegaconf -> attr-dict, hydra.utils.instantiate -> `_target_`
resolver, pytorch_lightning.LightningModule / torchmetrics.Metric -> thin nn.Module stand-ins, blanket stubs for
wandb / cv2 / tensorflow / waymo_open_dataset / h5py / transforms3d.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import copy
import importlib
import importlib.abc
import importlib.machinery
import inspect
import os
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

tb = load_package()
AttrDict, to_attr = tb.config.AttrDict, tb.config.to_attr


# --------------------------------------------------------------------------------------------- shims
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def _instantiate(cfg, *a, _recursive_=True, **kw):
    cfg = dict(cfg)
    mod, cls = cfg.pop("_target_").rsplit(".", 1)
    return getattr(importlib.import_module(mod), cls)(*a, **{**cfg, **kw})


class _LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.current_epoch, self.global_rank, self.logged = 0, 0, {}

    def save_hyperparameters(self):
        f = inspect.currentframe().f_back
        av = inspect.getargvalues(f)
        self.hparams = AttrDict({k: av.locals[k] for k in av.args if k != "self"})

    def log(self, k, v, **kw):
        self.logged[k] = v


class _Metric(nn.Module):
    def __init__(self):
        super().__init__()
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = default
        setattr(self, name, default.clone())

    def reset(self):
        for k, d in self._defaults.items():
            setattr(self, k, d.clone())

    def forward(self, *a, **k):
        self.update(*a, **k)
        return self.compute()


class _AnyMeta(type):
    def __getattr__(cls, k):
        return cls


class _Any(metaclass=_AnyMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("wandb", "cv2", "tensorflow", "waymo_open_dataset", "google", "h5py", "tqdm")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = lambda k: _Any
        return m

    def exec_module(self, module):
        pass


def install_shims():
    sys.modules["omegaconf"] = _module("omegaconf", DictConfig=AttrDict, ListConfig=list)
    sys.modules["transforms3d"] = _module("transforms3d", euler=SimpleNamespace(mat2euler=None, euler2mat=None))
    sys.modules["hydra"] = _module("hydra", utils=SimpleNamespace(instantiate=_instantiate, get_class=None))
    pl = _module("pytorch_lightning", LightningModule=_LightningModule)
    pl.loggers = _module("pytorch_lightning.loggers", WandbLogger=object)
    sys.modules["pytorch_lightning"], sys.modules["pytorch_lightning.loggers"] = pl, pl.loggers
    tm = _module("torchmetrics", Metric=_Metric)
    tm.metric = _module("torchmetrics.metric", Metric=_Metric)
    sys.modules["torchmetrics"], sys.modules["torchmetrics.metric"] = tm, tm.metric
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, str(REF))


# --------------------------------------------------------------------------------------------- helpers
def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(OUT / name, **out)
    print(f"wrote {name}: {sum(a.nbytes for a in out.values())/1e3:.1f} kB raw, {len(out)} arrays")


def ref_model_cfg(**kw):
    cfg = tb.config.default_model_cfg(**kw)
    return cfg


def ref_wm_kwargs(model_cfg, **sim_over):
    sim = tb.config.default_sim_cfg(**sim_over)
    model = copy.deepcopy(model_cfg)
    # these are supplied by WaymoMotion from pre_processing.model_kwargs / dynamics (waymo_motion.py:66-80)
    for k in ("tl_mode", "navi_mode", "navi_dim", "n_mp_pl_node", "mp_attr_dim", "tl_state_dim", "ag_motion_dim",
              "ag_attr_dim", "time_step_gt", "action_dim"):
        model.pop(k)
    model["_target_"] = "models.traffic_bots.TrafficBots"
    sim["model"] = model
    sim["pre_processing"]["scene_centric"]["_target_"] = "data_modules.scene_centric.SceneCentricPreProcessing"
    for k in ("veh", "cyc", "ped"):
        sim["dynamics"][k]["_target_"] = "utils.dynamics.MultiPathPP"
    sim["optimizer"]["_target_"] = "torch.optim.AdamW"
    sim["lr_scheduler"]["_target_"] = "torch.optim.lr_scheduler.StepLR"
    sim["data_size"] = to_attr(tb.synthetic.DATA_SIZE)
    sim.update(
        n_vis_batch=0, n_joint_future_womd=6, womd_post_processing=AttrDict(), wosac_post_processing=AttrDict(),
        sub_womd_reactive_replay=AttrDict(), sub_womd_joint_future_pred=AttrDict(), sub_wosac=AttrDict(),
    )
    return sim


def build_wm(model_cfg, **sim_over):
    import pl_modules.waymo_motion as wm

    for n in ("WOSACMetrics", "WOMDMetrics", "SubWOMD", "SubWOSAC", "WOSACPostProcessing", "WOMDPostProcessing",
              "ErrorMetrics", "TrafficRuleMetrics"):
        setattr(wm, n, lambda *a, **k: None)
    return wm.WaymoMotion(**ref_wm_kwargs(model_cfg, **sim_over))


def sorted_valid_sets(idx, invalid):
    """KNN parity object (SURVEY Appx A.2): per row, sorted indices of the non-masked neighbours, -1 padded."""
    idx = idx.clone()
    idx[invalid] = 2**31 - 1
    idx = idx.sort(-1)[0]
    idx[idx == 2**31 - 1] = -1
    return idx


# --------------------------------------------------------------------------------------------- fixtures
@torch.no_grad()
def gen_ops():
    """§8a rows 1-6: op-level vectors (seeded inputs, det_fill weights)."""
    from models.modules.attention_rpe import AttentionRPE
    from models.modules.input_encoder import InputEncoder
    from models.modules.mlp import MLP
    from models.modules.polyline_encoder import PolylineEncoder
    from models.modules.transformer_rpe import TransformerBlockRPE
    from utils.pooling import seq_pooling
    from utils.pose_emb import PoseEmb
    from utils.rpe import get_rel_pose, get_tgt_knn_idx

    g = torch.Generator().manual_seed(1234)
    out = {}
    # rows 1,2: relative pose + KNN
    n, S, T, K = 2, 12, 40, 6
    pose = torch.cat([(torch.rand(n, S, 2, generator=g) - 0.5) * 200, (torch.rand(n, S, 1, generator=g) - 0.5) * 6.28], -1)
    pose2 = torch.cat([(torch.rand(n, T, 2, generator=g) - 0.5) * 200, (torch.rand(n, T, 1, generator=g) - 0.5) * 6.28], -1)
    inv = torch.rand(n, S, generator=g) < 0.2
    inv2 = torch.rand(n, T, generator=g) < 0.3
    inv2[1, 3:] = True  # fewer than K valid targets in scene 1
    rel_pose, rel_dist = get_rel_pose(pose, inv, pose2, inv2)
    idx, knn_inv, rpe = get_tgt_knn_idx(inv2, rel_pose, rel_dist, K, 80.0)
    out.update(rel_pose=rel_pose, rel_dist=rel_dist, knn_sets=sorted_valid_sets(idx, knn_inv), knn_n_valid=(~knn_inv).sum(-1))
    rel_pose_s, rel_dist_s = get_rel_pose(pose, inv)
    idx_s, knn_inv_s, _ = get_tgt_knn_idx(inv, rel_pose_s, rel_dist_s, 5, 150.0)
    out.update(rel_dist_self=rel_dist_s, knn_sets_self=sorted_valid_sets(idx_s, knn_inv_s))
    # integer-lattice case: every distance exact in fp32
    lat = torch.cat([torch.randint(-40, 40, (1, 30, 2), generator=g).float() * 0.25,
                     torch.randint(0, 4, (1, 30, 1), generator=g).float() * (np.pi / 2)], -1)
    lat_inv = torch.zeros(1, 30, dtype=torch.bool)
    rp, rd = get_rel_pose(lat, lat_inv)
    out.update(lattice_rel_dist=rd)
    # row 3: pose embeddings
    for dim in (128, 64):
        pe = PoseEmb("pe_xy_yaw", pe_dim=dim, theta_xy=1e3)
        out[f"pe_xy_yaw_{dim}"] = pe(rel_pose[..., :2], rel_pose[..., 2:3])
    out["mpa_pl"] = PoseEmb("mpa_pl").forward(rel_pose[..., :2], rel_pose[..., 2:3])
    # row 4: pooling
    x = torch.randn(2, 5, 7, 16, generator=g)
    xi = torch.rand(2, 5, 7, generator=g) < 0.4
    xi[0, 0] = True
    for mode in ("max_valid", "last_valid", "mean_valid", "first", "last"):
        out[f"pool_{mode}"] = seq_pooling(x, xi, mode)
    # row 5: MLP / InputEncoder / PointNet
    mlp = MLP([20, 64, 64, 64], end_layer_activation=False).eval()
    tb.utils.det_fill(mlp, 11)
    xa = torch.randn(2, 5, 7, 20, generator=g)
    out["mlp"] = mlp(xa)
    out["mlp_masked"] = mlp(xa, xi, float("-inf"))
    mlp_ln = MLP([48, 32, 32, 1], end_layer_activation=False, use_layernorm=True).eval()
    tb.utils.det_fill(mlp_ln, 12)
    out["mlp_ln"] = mlp_ln(torch.randn(3, 48, generator=g))
    for mode, pe_dim in (("cat", 64), ("add", 128)):
        ie = InputEncoder(128, 20, pe_dim, 3, 0, False, mode).eval()
        tb.utils.det_fill(ie, 13)
        out[f"input_encoder_{mode}"] = ie(xa, torch.randn(2, 5, 7, pe_dim, generator=g))
    pn = PolylineEncoder(128, AttrDict(), 3, False, 0.1, True, "max_valid").eval()
    tb.utils.det_fill(pn, 14)
    out["pointnet"] = pn(torch.randn(2, 5, 7, 128, generator=g), xi)
    # row 6: AttentionRPE (rpe branch), mixed + all-invalid rows
    d = 128
    att = AttentionRPE(d, 4, dropout_p=0.1, d_rpe=d).eval()
    tb.utils.det_fill(att, 15)
    n, S, K = 2, 9, 11
    src = torch.randn(n, S, d, generator=g)
    tgt = torch.randn(n, S, K, d, generator=g)
    rpe_e = torch.randn(n, S, K, d, generator=g)
    m = torch.rand(n, S, K, generator=g) < 0.3
    m[0, 2] = True
    m[1, 0] = True
    out["attn_rpe"] = att(src, tgt, tgt_padding_mask=m, rpe=rpe_e)[0]
    # row 7: one TransformerBlockRPE per mode used by the default model
    tf_cfg = dict(d_model=d, n_head=4, k_feedforward=4, dropout_p=0.1, bias=True, activation="relu",
                  out_layernorm=False, apply_q_rpe=False)
    src_inv = torch.rand(n, S, generator=g) < 0.2
    idx_self = torch.randint(0, S, (n, S, 5), generator=g)
    m_self = torch.rand(n, S, 5, generator=g) < 0.3
    rpe_self = torch.randn(n, S, 5, d, generator=g)
    enc = TransformerBlockRPE(n_layer=2, mode="enc_self_attn", d_rpe=d, **tf_cfg).eval()
    tb.utils.det_fill(enc, 16)
    out["tf_enc_self"] = enc(src=src.clone(), src_padding_mask=src_inv, tgt=idx_self, tgt_padding_mask=m_self, rpe=rpe_self)[0]
    dec = TransformerBlockRPE(n_layer=2, mode="dec_cross_attn", d_rpe=d, **tf_cfg).eval()
    tb.utils.det_fill(dec, 17)
    out["tf_dec_cross"] = dec(src=src.clone(), src_padding_mask=src_inv, tgt=tgt, tgt_padding_mask=m, rpe=rpe_e,
                              decoder_tgt=idx_self, decoder_tgt_padding_mask=m_self, decoder_rpe=rpe_self)[0]
    npz("ops.npz", **out)


def _to_dev(d):
    return {k: v for k, v in d.items()}


def gen_model(tag, n_ag, n_mp, n_tl, n_tgt_knn, n_roll_steps, train_fixture):
    """§8a rows 8-20 at one size: tokens, per-step heads, closed-loop rollout, training step."""
    torch.manual_seed(0)
    mcfg = ref_model_cfg(n_tgt_knn=n_tgt_knn)
    wm = build_wm(mcfg)
    tb.utils.det_fill(wm.model, 0)
    n_par = sum(p.numel() for p in wm.model.parameters())
    keys = sorted(wm.model.state_dict().keys())
    batch = tb.synthetic.make_scene(1, n_ag, n_mp, n_tl, seed=0)
    out = {"n_params": np.int64(n_par)}
    if tag == "c1":
        (OUT / "state_dict_keys.txt").write_text(
            "\n".join(f"{k} {tuple(wm.model.state_dict()[k].shape)}" for k in keys) + "\n")

    # ---- eval: once-per-scene tokens + reactive replay (rows 8,9,10,12-18,20)
    wm.eval()
    with torch.no_grad():
        # validation batches carry the full episode and its `history/*` view (data_h5_womd.py tensor_size_val)
        b = wm.pre_processing({k: v.clone() for k, v in {**batch, **tb.synthetic.to_history_batch(batch)}.items()})
        mp_tokens = wm.model.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_tokens = wm.model.tl_encoder.pre_compute(
            tl_valid=b["gt/tl_valid"], tl_attr=b["sc/tl_attr"], tl_pose=b["sc/tl_pose"], **mp_tokens)
        feat = mp_tokens["mp_token_feature"]
        out.update(mp_token_feature_head=feat[:, :32], mp_token_feature_sum=feat.double().sum(),
                   mp_token_feature_abs=feat.double().abs().sum(), mp_token_invalid=mp_tokens["mp_token_invalid"])
        out.update(
            tl2tl_sets=sorted_valid_sets(tl_tokens["knn_idx_tl2tl"], tl_tokens["knn_invalid_tl2tl"]),
            tl2mp_n_valid=(~tl_tokens["knn_invalid_tl2mp"]).sum(-1),
            tl_token_attr=tl_tokens["tl_token_attr"][:, :16],
        )
        latent_post = wm.model.latent_encoder(
            ag_valid=b["gt/ag_valid"], ag_attr=b["sc/ag_attr"], ag_motion=b["gt/ag_motion"], ag_pose=b["gt/ag_pose"],
            ag_type=b["ref/ag_type"], tl_state=b["gt/tl_state"], mp_tokens=mp_tokens, tl_tokens=tl_tokens, posterior=True)
        out.update(latent_post_mean=latent_post.mean, latent_post_valid=latent_post.valid)
        navi_pred = wm.model.navi_predictor(
            ag_valid=b["sc/ag_valid"], ag_attr=b["sc/ag_attr"], ag_motion=b["sc/ag_motion"], ag_pose=b["sc/ag_pose"],
            ag_type=b["ref/ag_type"], **mp_tokens)
        out.update(navi_log_prob_gt=navi_pred.log_prob(b["gt/ag_navi"]), navi_valid=navi_pred.valid,
                   navi_argmax=navi_pred.probs.argmax(-1))
        # closed-loop replay with the reactive_replay teacher forcing, posterior-mean latent, GT dest
        wm.hparams.time_step_end = n_roll_steps
        buf = wm.reactive_replay(
            batch=b, mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_latent=latent_post.sample(deterministic=True),
            ag_latent_valid=latent_post.valid, ag_navi=b["gt/ag_navi"], ag_navi_valid=b["gt/ag_valid"].any(-1),
            teacher_forcing=wm.teacher_forcing_joint_future_pred, deterministic_action=True)
        wm.hparams.time_step_end = 90
        out.update(
            rr_pred_valid=buf.pred_valid[:, 0], rr_pred_pose=buf.pred_pose[:, 0], rr_pred_motion=buf.pred_motion[:, 0],
            rr_tl_state_nll=buf.tl_state_nll[:, 0], rr_outside_map=buf.violation["outside_map"][:, 0],
            rr_dest_reached=buf.violation["dest_reached"][:, 0], rr_action=buf.vis_dict["action"][:, 0],
            rr_tl_state=buf.vis_dict["tl_state"][:, 0], rr_diffbar_reward=buf.diffbar_reward["diffbar_reward"][:, 0],
        )

    # ---- train: training_step loss + per-module grad norms, all RNG sites neutralised (row 19,20)
    if train_fixture:
        mcfg0 = ref_model_cfg(n_tgt_knn=n_tgt_knn)
        mcfg0["tf_cfg"]["dropout_p"] = 0.0
        mcfg0["mp_encoder"]["pl_encoder"]["mlp_dropout_p"] = 0.0
        mcfg0["add_navi_latent"]["mlp_dropout_p"] = 0.0
        wm_t = build_wm(mcfg0, p_training_rollout_prior=0.0)
        wm_t.hparams.teacher_forcing_training  # noqa: B018
        wm_t.teacher_forcing_training.prob_forcing_agent = 0.0
        wm_t.pre_processing[0].dropout_p_history = -1.0
        tb.utils.det_fill(wm_t.model, 0)
        wm_t.train()
        torch.manual_seed(7)
        loss = wm_t.training_step({k: v.clone() for k, v in batch.items()}, 0)
        loss.backward()
        out["train_loss"] = loss.detach()
        for k, v in wm_t.logged.items():
            out["train_" + k.split("/")[1]] = v.detach()
        gn = {}
        for k, p in wm_t.model.named_parameters():
            top = k.split(".")[0]
            if p.grad is not None:
                gn[top] = gn.get(top, 0.0) + float(p.grad.double().pow(2).sum())
        for k, v in gn.items():
            out["gradnorm_" + k] = np.float64(v) ** 0.5
        no_grad = sorted(k for k, p in wm_t.model.named_parameters() if p.grad is None)
        (OUT / "params_without_grad.txt").write_text("\n".join(no_grad) + "\n")
        # same step with the action head's output layer scaled by 0.02 ("damped"): the closed loop is then not chaotic,
        # so implementations with a different fp32 summation order can be compared tightly on loss AND gradients
        wm_t.zero_grad(set_to_none=True)
        with torch.no_grad():
            for k, p in wm_t.model.named_parameters():
                if k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k:
                    p.mul_(0.02)
        torch.manual_seed(7)
        loss = wm_t.training_step({k: v.clone() for k, v in batch.items()}, 0)
        loss.backward()
        out["dtrain_loss"] = loss.detach()
        for k, v in wm_t.logged.items():
            out["dtrain_" + k.split("/")[1]] = v.detach()
        gn = {}
        for k, p in wm_t.model.named_parameters():
            top = k.split(".")[0]
            if p.grad is not None:
                gn[top] = gn.get(top, 0.0) + float(p.grad.double().pow(2).sum())
        for k, v in gn.items():
            out["dgradnorm_" + k] = np.float64(v) ** 0.5
        for k in ("ag_encoder.tf_ag2agmptl.layers.3.attn.linear_rpe.weight", "mp_encoder.tf_mp2mp.layers.0.attn.in_proj_weight",
                  "tl_encoder.tf_tl2tlmp.layers.1.attn_src.out_proj_weight", "latent_encoder.ag_encoder_post.input_encoder.mlp.fc_layers.0.weight"):
            out["dgrad_" + k] = dict(wm_t.model.named_parameters())[k].grad[:8, :16]
    npz(f"model_{tag}.npz", **out)


def gen_train(tag, n_ag, n_mp, n_tl, n_tgt_knn, n_sc=1, edge=False):
    """§8a rows 19-20 at the size of BASELINE config 3's scenes (one scene of 64 agents / 1024 polylines / 128 lights): the
    reference's training_step with every RNG site neutralised and the damped action head (see gen_model): loss dict, per-module
    gradient norms, spot gradients. ~3 min of reference CPU time. n_sc > 1: a BATCH of scenes (BASELINE config 3 trains on batches:
    the loss terms are ratios of sums over the whole batch, metrics/training.py:166-186)."""
    torch.manual_seed(0)
    mcfg0 = ref_model_cfg(n_tgt_knn=n_tgt_knn)
    mcfg0["tf_cfg"]["dropout_p"] = 0.0
    mcfg0["mp_encoder"]["pl_encoder"]["mlp_dropout_p"] = 0.0
    mcfg0["add_navi_latent"]["mlp_dropout_p"] = 0.0
    wm_t = build_wm(mcfg0, p_training_rollout_prior=0.0)
    wm_t.teacher_forcing_training.prob_forcing_agent = 0.0
    wm_t.pre_processing[0].dropout_p_history = -1.0
    tb.utils.det_fill(wm_t.model, 0)
    wm_t.train()
    with torch.no_grad():
        for k, p in wm_t.model.named_parameters():
            if k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k:
                p.mul_(0.02)
    # edge: three scenes with the domain's empty inputs - no valid light / two agents / no valid polyline (synthetic.make_edge_batch)
    # edge = "no_lights": one scene without a valid light (the light-state term's counter is zero: metrics/training.py:184 leaves it out)
    batch = (tb.synthetic.make_edge_batch(n_ag, n_mp, n_tl, seed=0, kind=edge if isinstance(edge, str) else "mixed") if edge
             else tb.synthetic.make_scene(n_sc, n_ag, n_mp, n_tl, seed=0))
    torch.manual_seed(7)
    loss = wm_t.training_step({k: v.clone() for k, v in batch.items()}, 0)
    loss.backward()
    out = {"dtrain_loss": loss.detach()}
    for k, v in wm_t.logged.items():
        out["dtrain_" + k.split("/")[1]] = v.detach()
    gn = {}
    for k, p in wm_t.model.named_parameters():
        top = k.split(".")[0]
        if p.grad is not None:
            gn[top] = gn.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in gn.items():
        out["dgradnorm_" + k] = np.float64(v) ** 0.5
    for k in ("ag_encoder.tf_ag2agmptl.layers.3.attn.linear_rpe.weight", "mp_encoder.tf_mp2mp.layers.0.attn.in_proj_weight",
              "tl_encoder.tf_tl2tlmp.layers.1.attn_src.out_proj_weight", "latent_encoder.ag_encoder_post.input_encoder.mlp.fc_layers.0.weight",
              "navi_predictor.mlp.fc_layers.0.weight", "action_head.mlp_mean.0.fc_layers.0.weight"):
        out["dgrad_" + k] = dict(wm_t.model.named_parameters())[k].grad[:8, :16]
    npz(f"train_{tag}.npz", **out)


@torch.no_grad()
def gen_rules():
    """the reference's TrafficRuleChecker stepped over seeded crowded episodes. Inputs come from
    `synthetic.make_rule_episode(seed)` (ours, shipped); the fixture holds the reference's outputs only."""
    from utils.traffic_rule_checker import TrafficRuleChecker
    from utils.wosac_collision import check_collided_wosac

    out = {}
    for tag, kw in (("a", dict(n_sc=2, n_ag=16, n_mp=64, n_tl=8, n_step=40, seed=0)),
                    ("b", dict(n_sc=1, n_ag=40, n_mp=96, n_tl=12, n_step=30, seed=7, extent=40.0))):
        e = tb.synthetic.make_rule_episode(**kw)
        rc = TrafficRuleChecker(
            mp_boundary=e["map/boundary"], mp_valid=e["map/valid"], mp_type=e["map/type"], mp_pos=e["map/pos"], mp_dir=e["map/dir"],
            ag_type=e["agent/type"], ag_size=e["agent/size"], ag_goal=None, ag_dest=None, tl_valid=e["tl/valid"],
            tl_pose=e["tl/pose"], disable_check=False)
        T = e["agent/valid"].shape[2]
        log = {}
        for t in range(T):
            v = rc.check(e["agent/valid"][:, :, t], e["agent/pose"][:, :, t], e["agent/motion"][:, :, t], e["tl/state"][:, :, t])
            for k, x in v.items():
                log.setdefault(k, []).append(x.clone())
        for k in ("collided", "collided_wosac", "run_road_edge", "run_red_light", "passive"):
            out[f"{tag}_{k}"] = np.packbits(torch.stack(log[k], 2).numpy(), axis=-1)
            out[f"{tag}_{k}_this_step"] = np.packbits(torch.stack(log[k + "_this_step"], 2).numpy(), axis=-1)
            print(tag, k, int(torch.stack(log[k + "_this_step"], 2).sum()), "this-step positives")
        out[f"{tag}_passive_counter"] = rc.passive_counter
        out[f"{tag}_n_step"] = np.int64(T)
    npz("rules.npz", **out)


@torch.no_grad()
def gen_filter():
    """the reference's WOSACPostProcessing._filter_futures on seeded inputs
    (`synthetic.make_filter_case`, ours). Kept: the per-rollout violation score it ranks by (recomputed with the
    reference's expressions from its own inputs) and the SORTED scores / indices of the futures it keeps."""
    from data_modules.wosac_post_processing import WOSACPostProcessing
    from utils.buffer import RolloutBuffer

    out = {}
    for tag, kw, w, wosac in (("a", dict(n_sc=2, n_k=48, n_ag=12, n_step=30, seed=0), 0.5, True),
                              ("b", dict(n_sc=3, n_k=128, n_ag=20, n_step=91, seed=1, p_col=0.05, p_edge=0.1), 2.0, False)):
        c = tb.synthetic.make_filter_case(**kw)
        pp = WOSACPostProcessing(step_gt=90, step_current=10, const_vel_z_sim=True, const_vel_no_sim=True, w_road_edge=w,
                                 use_wosac_col=wosac)
        buf = RolloutBuffer(c["pred_pose"].shape[3], 10)
        buf.pred_pose = c["pred_pose"]
        buf.violation = {k: c[k] for k in ("collided", "collided_wosac", "run_road_edge")}
        trajs = pp._filter_futures(buf, c["ag_role"])
        # which futures were kept: match the returned trajectories back to their rollout index
        flat = c["pred_pose"][:, :, :, buf.step_future_start:]
        idx = torch.stack([torch.stack([torch.nonzero((flat[s] == trajs[s, j]).flatten(1).all(1))[0, 0] for j in range(trajs.shape[1])])
                           for s in range(trajs.shape[0])])
        out[f"{tag}_idx_sorted"] = idx.sort(-1)[0]
        out[f"{tag}_trajs_checksum"] = trajs.double().sum((1, 2, 3, 4))
        print(tag, "kept", trajs.shape, "first scene idx", out[f"{tag}_idx_sorted"][0][:8].tolist())
    npz("filter.npz", **out)


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    # (train_c2_b4 - a batch of 4 full-size scenes, ~12 min of reference CPU and ~40 GB of autograd graph - only on request)
    which = sys.argv[1:] or ["ops", "c1", "c2", "train_c2", "train_c1_b3", "train_c1_edge", "train_c1_nolights", "rules", "filter"]
    if "filter" in which:
        gen_filter()
    if "rules" in which:
        gen_rules()
    if "ops" in which:
        gen_ops()
    if "c1" in which:
        gen_model("c1", 8, 64, 8, 4, n_roll_steps=90, train_fixture=True)
    if "c2" in which:
        gen_model("c2", 64, 1024, 128, 32, n_roll_steps=14, train_fixture=False)
    if "train_c2" in which:
        gen_train("c2", 64, 1024, 128, 32)
    if "train_c2_b4" in which:
        gen_train("c2_b4", 64, 1024, 128, 32, n_sc=4)
    if "train_c1_b3" in which:
        gen_train("c1_b3", 8, 64, 8, 4, n_sc=3)
    if "train_c1_edge" in which:
        gen_train("c1_edge", 8, 64, 8, 4, n_sc=3, edge=True)
    if "train_c1_nolights" in which:
        gen_train("c1_nolights", 8, 64, 8, 4, n_sc=1, edge="no_lights")
