"""Differentiable (training) schedule of the hot path on the GPU.

Forward AND backward of the KNARPE attention are hand-written HIP kernels (tbx_knarpe_attn_fwd / _bwd) behind
`KnarpeAttnFn`; K-nearest selection, pose embeddings and feature preparation are the same HIP kernels as in inference
(no gradient flows through them: the reference computes them under no_grad, utils/rpe.py:7,40,61). The dense
projections are plain library GEMMs (`F.linear` -> hipBLASLt) and the LayerNorm / ReLU / masking / max-pool glue is
elementwise torch on the device, so autograd provides their backward. Fusing those into chain-backward kernels is the
next step; the formulation (K/V projected before the gather, linear_rpe folded) is identical to the
inference engine, so both are checked against the same oracle.

Everything takes the reference-named nn.Modules as parameter containers (same state dict as inference).
This module assembles the encoders, the policy step, the two rollout forms, the loss and `training_step`; the autograd Functions
and elementwise / LINEAR helpers live in train_ops.py, the shared state (arithmetic class, dropout scope, per-step caches) in
train_state.py.

Time-batched rollout (`training_rollout_batched`, the default): the reference detaches the policy inputs of every
closed-loop step (waymo_motion.py:206-311 with training=True), so the only gradient path across steps is the dynamics
chain. The rollout therefore runs twice: (1) step by step WITHOUT autograd, recording each step's (detached) policy inputs;
(2) the 90 policy evaluations of a scene as ONE batch of 90 x n_scene entries WITH autograd (map K/V tables shared through
`batch_div`, dropout masks keyed by (site, step, row, column) so that the batched pass draws the masks of the sequential
one), followed by the tiny per-step dynamics / reward chain on the batched means. Same loss and gradients as stepping with
autograd (`training_rollout`, kept as the checked reference of the restructure), but ~25 large launches per layer
instead of 90 x as many small ones.
"""
import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.distributions import Categorical, Independent, Normal, kl_divergence

from . import hip, hip_base
from . import train_state as ST
from . import train_state as state  # noqa: F401  (tests switch paths / classes through train_graph.state.X)
from .hip import Seg
from .models.modules.distributions import DestCategorical, DiagGaussian
from .train_ops import *  # noqa: F401,F403  (autograd Functions + linear / layer_norm / residual / mlp / pointnet / attention helpers)
from .train_ops import _attn_residual, _drop, _drop_args, _glue_ok, _pointnet_fused_ok  # noqa: F401
from .train_state import (ATTN_MFMA_MIN_ROWS, HEADS_TILE, TALL_LINEAR, WGRAD_MIN_ROWS, _DropScope, _POLICY_SITE0, bf16_contractions,  # noqa: F401
                          module_scope, precision)

D, NH, DH = 128, 4, 32


def _chains_ok(x: Tensor) -> bool:
    return ST.NOGRAD_CHAINS and not torch.is_grad_enabled() and x.is_cuda and (ST._DROP is None or ST._DROP["tb"] == 1)


def _transformer_block_chains(block, x, src_invalid, n, S, self_knn, cross, p, training) -> Tensor:
    from .engine import SelfKnn, kv_tables, run_block

    layers = list(block.layers)
    xx = x.contiguous().clone()
    knn = SelfKnn(self_knn["idx"], self_knn["invalid"], emb=self_knn.get("emb"), rel=self_knn.get("rel"))
    cross_fn = None
    if block.mode == "dec_cross_attn":
        tg = list(cross(layers[0]))  # the same token sets for every layer
        tabs = []
        for t in tg:  # K|V tables of all layers in one chain launch ([tokens, n_layer * 256]); static sets: once per training step
            ck = None if (t.cache is None or t.key is None) else (t.key, id(block), "all-layers")
            if ck is None or ck not in t.cache:
                tab = kv_tables(t.tokens.contiguous(), [(l.norm_tgt, l.attn) for l in layers])
                if ck is not None:
                    t.cache[ck] = tab
            tabs.append(tab if ck is None else t.cache[ck])
        cross_fn = lambda l: [Seg(tab, l * 2 * D, l * 2 * D + D, t.n_tgt, t.idx, t.invalid, t.emb, t.batch_div, rel=t.rel)
                              for tab, t in zip(tabs, tg)]
    drop = None
    if training and ST._DROP is not None:
        drop = dict(p=float(p), seed=ST._DROP["seed"], site=ST._DROP["site"], call=ST._DROP["call"], step=ST._DROP["t0"])
    run_block(block, xx, src_invalid, n, S, knn, cross=cross_fn, freqs=self_knn.get("freqs"), drop=drop)
    if drop is not None:
        ST._DROP["site"], ST._DROP["call"] = drop["site"], drop["call"]
    return xx


def transformer_block(block, x: Tensor, src_invalid: Tensor, n: int, S: int, self_knn, cross=None, p: float = 0.0,
                      training: bool = False) -> Tensor:
    """transformer_rpe.py:48-135,207-245. x [n*S,128]; self_knn = Targets kwargs (idx, invalid, emb | rel, freqs) among the sources;
    cross(layer) -> list[Targets] with UN-normalised tokens (norm_tgt is applied here)."""
    if _chains_ok(x):
        return _transformer_block_chains(block, x, src_invalid, n, S, self_knn, cross, p, training)
    ln = lambda m, t: layer_norm_join(t, m)  # -> (x to continue the residual branch from, LayerNorm(x)): train_ops.LayerNormJoinFn
    inv = src_invalid.reshape(-1).to(torch.uint8)
    # the glue between the GEMMs / attention calls (zeroing of rows without a valid target, dropout, residual add, relu, the closing
    # row mask) as one pass per tensor: residual() / relu_drop() - the dropout sites keep the order of the reference's modules
    for layer in block.layers:
        if block.mode == "dec_cross_attn":
            x, s = ln(layer.norm_src, x)
            ts = Targets(s, n_tgt=S, **self_knn)
            x = _attn_residual(x, attention(layer.attn_src, s, [ts], [kv_table(layer.attn_src, None, ts)], n, S, raw=True), p, training)
            x, s2 = ln(layer.norm1, x)
            tg = list(cross(layer))
            x = _attn_residual(x, attention(layer.attn, s2, tg, [kv_table(layer.attn, layer.norm_tgt, t) for t in tg], n, S, raw=True), p, training)
        else:  # enc_self_attn: gathered targets share norm1 with the source
            x, s2 = ln(layer.norm1, x)
            ts = Targets(s2, n_tgt=S, **self_knn)
            x = _attn_residual(x, attention(layer.attn, s2, [ts], [kv_table(layer.attn, None, ts)], n, S, raw=True), p, training)
        x, s3 = ln(layer.norm2, x)
        h = linear_relu_drop(s3, layer.linear1.weight, layer.linear1.bias, p, training)
        x = residual(x, linear(h, layer.linear2.weight, layer.linear2.bias), p, training, zero_out=inv)
    return x


def _knn(src_pose, src_inv, tgt_pose, tgt_inv, k, limit, rp, div=1):
    """-> kwargs of Targets: KNN indices / mask + the relative poses (the embedding is rebuilt inside the attention kernels)."""
    idx, inv, rel, _ = hip.knn_embed(src_pose, src_inv, tgt_pose, tgt_inv, k, limit, tgt_batch_div=div, want_rel_pose=True,
                                     want_emb=False)
    # with gradients on: the inverse lists the attention backward gathers dK / dV through
    lists = hip.knn_inverse(idx, inv, tgt_pose.shape[1], div) if torch.is_grad_enabled() else None
    return dict(idx=idx, invalid=inv, emb=None, rel=rel, freqs=(rp.pe_xy.freqs, rp.pe_yaw.freqs), inv=lists)


# ------------------------------------------------------------------------------------------------ encoders
def map_encoder(me, mp_valid, mp_attr, mp_pose, mp_type, training: bool) -> Dict[str, Tensor]:
    n, M, N = mp_valid.shape
    dev, d = mp_pose.device, me.hidden_dim
    rows = n * M * N
    attr = torch.empty(rows, 32, dtype=torch.float32, device=dev)
    pe = torch.empty(rows, 8, dtype=torch.float32, device=dev)
    row_inv = torch.empty(rows, dtype=torch.uint8, device=dev)
    tok_pose = torch.empty(n, M, 3, dtype=torch.float32, device=dev)
    tok_inv = torch.empty(n, M, dtype=torch.uint8, device=dev)
    hip.map_prep(mp_valid.to(torch.uint8).contiguous(), mp_attr.float().contiguous(), mp_pose.float().contiguous(), attr, pe, row_inv,
                 tok_pose, tok_inv)
    x = torch.cat([mlp(me.input_encoder.mlp, attr[:, :me.input_encoder.mlp.input_dim], training), pe[:, :7]], -1)
    feat = pointnet(me.pl_encoder, x.view(n * M, N, d), row_inv.view(n * M, N).bool(), training)
    knn = _knn(tok_pose, tok_inv, tok_pose, tok_inv, me.n_tgt_knn, me.dist_limit, me.pose_rpe)
    feat = transformer_block(me.tf_mp2mp, feat, tok_inv, n, M, knn, p=me.tf_mp2mp.dropout_p, training=training)
    return {"mp_token_invalid": tok_inv.bool(), "mp_token_invalid_u8": tok_inv, "mp_token_feature": feat.view(n, M, d),
            "mp_token_pose": tok_pose, "mp_token_type": mp_type}


def tl_pre_compute(te, tl_valid, tl_attr, tl_pose, mp: Dict[str, Tensor]) -> Dict[str, Tensor]:
    n, L = tl_valid.shape
    dev = tl_pose.device
    tl_inv = (~tl_valid).to(torch.uint8).contiguous()
    pose = tl_pose.float().contiguous()
    mpf = mp["mp_token_feature"].detach() if te.tl_lane_detach_mp_feature else mp["mp_token_feature"]
    t = {"tl_token_valid": tl_valid, "tl_token_invalid": ~tl_valid, "tl_token_invalid_u8": tl_inv, "tl_token_pose": pose,
         "tl_token_attr": mpf[torch.arange(n, device=dev).unsqueeze(1), tl_attr], "mp_feat_for_tl": mpf}
    t["tt"] = _knn(pose, tl_inv, pose, tl_inv, te.n_tgt_knn_tl2tl, te.dist_limit, te.pose_rpe)
    t["tm"] = _knn(pose, tl_inv, mp["mp_token_pose"], mp["mp_token_invalid_u8"], te.n_tgt_knn_tl2mp, te.dist_limit, te.pose_rpe)
    return t


def expand_tl_tokens(te, t: Dict[str, Tensor], mp: Dict[str, Tensor], T: int) -> Dict[str, Tensor]:
    """tl_pre_compute's per-scene dict for a time-batched call: batch entry b = scene b // T. Per-light tensors are repeated,
    the K-nearest sets are searched again on the repeated poses (identical sets; the map tables stay per scene: div = T)."""
    rep = lambda x: x.repeat_interleave(T, 0)
    e = {k: rep(t[k]) for k in ("tl_token_valid", "tl_token_invalid", "tl_token_invalid_u8", "tl_token_pose", "tl_token_attr")}
    e["tl_token_invalid_u8"], e["tl_token_pose"] = e["tl_token_invalid_u8"].contiguous(), e["tl_token_pose"].contiguous()
    e["mp_feat_for_tl"], e["time_batch"] = t["mp_feat_for_tl"], T
    e["tt"] = _knn(e["tl_token_pose"], e["tl_token_invalid_u8"], e["tl_token_pose"], e["tl_token_invalid_u8"], te.n_tgt_knn_tl2tl,
                   te.dist_limit, te.pose_rpe)
    e["tm"] = _knn(e["tl_token_pose"], e["tl_token_invalid_u8"], mp["mp_token_pose"], mp["mp_token_invalid_u8"], te.n_tgt_knn_tl2mp,
                   te.dist_limit, te.pose_rpe, div=T)
    e["_kv_cache"] = {}
    return e


def tl_encoder(te, hist_tl: Tensor, t: Dict[str, Tensor], training: bool) -> Tensor:
    """hist_tl [n,L,W] u8 masks (0xFF missing) -> [n*L, 128]. traffic_light.py:184-246. `t` = tl_pre_compute's dict, or
    expand_tl_tokens' (n = scenes x time batch, the map tokens shared by the T entries of a scene)."""
    n, L, W = hist_tl.shape
    dev, d = hist_tl.device, te.hidden_dim
    ld = 16 if 5 + W <= 16 else 32
    attr = torch.empty(n * L * W, ld, dtype=torch.float32, device=dev)
    row_inv = torch.empty(n * L * W, dtype=torch.uint8, device=dev)
    hip.tl_prep(hist_tl, t["tl_token_invalid_u8"], attr, row_inv)
    x = mlp(te.input_encoder.mlp, attr[:, :5 + W], training).view(n * L, W, d) + t["tl_token_attr"].reshape(n * L, 1, d)
    x = pointnet(te.temp_encoder, x, row_inv.view(n * L, W).bool(), training)
    M = t["mp_feat_for_tl"].shape[1]
    mp_tokens = t["mp_feat_for_tl"].reshape(-1, d)
    return transformer_block(te.tf_tl2tlmp, x, t["tl_token_invalid_u8"], n, L, t["tt"],
                             cross=lambda layer: [Targets(mp_tokens, n_tgt=M, cache=t.get("_kv_cache"), key="tl2mp",
                                                          batch_div=t.get("time_batch", 1), **t["tm"])],
                             p=te.tf_tl2tlmp.dropout_p, training=training)


def agent_encoder(ae, hist_valid, hist_pose, hist_motion, ag_attr6, mp, tl_inv_u8, tl_pose, tl_feat, training: bool, T: int = 1):
    """agent_encoder.py:114-178,321-387. hist_* [n,A,W(,3)] oldest first -> (feat [n*A,128], prep dict). T > 1: time-batched
    call, n = scenes x T, `mp` per scene (shared by a scene's T entries), the light tensors per entry."""
    n, A, W = hist_valid.shape
    dev, d = hist_pose.device, ae.hidden_dim
    prep = ae.alloc_prep(n, A, dev, with_heads=False)
    hip.agent_prep(hist_valid, hist_pose, hist_motion, ag_attr6, None, ae.pose_emb.pe_xy.freqs, ae.pose_emb.pe_yaw.freqs,
                   ae.pose_emb.out_dim, prep)
    tok_pose, tok_inv = prep["tok_pose"], prep["tok_invalid"]
    aa = _knn(tok_pose, tok_inv, tok_pose, tok_inv, ae.n_tgt_knn_ag2ag, ae.dist_limit, ae.pose_rpe)
    am = _knn(tok_pose, tok_inv, mp["mp_token_pose"], mp["mp_token_invalid_u8"], ae.n_tgt_knn_ag2mp, ae.dist_limit, ae.pose_rpe, div=T)
    at = _knn(tok_pose, tok_inv, tl_pose, tl_inv_u8, ae.n_tgt_knn_ag2tl, ae.dist_limit, ae.pose_rpe)
    x = torch.cat([mlp(ae.input_encoder.mlp, prep["attr"][:, :ae.input_encoder.mlp.input_dim], training), prep["pe"]], -1)
    x = pointnet(ae.temp_encoder, x.view(n * A, W, d), prep["row_invalid"].view(n * A, W).bool(), training)
    M, L = mp["mp_token_pose"].shape[1], tl_pose.shape[1]
    mp_tokens = mp["mp_token_feature"].reshape(-1, d)
    x = transformer_block(ae.tf_ag2agmptl, x, tok_inv, n, A, aa,
                          cross=lambda layer: [Targets(mp_tokens, n_tgt=M, cache=mp.get("_kv_cache"), key="ag2mp", batch_div=T, **am),
                                               Targets(tl_feat, n_tgt=L, **at)],
                          p=ae.tf_ag2agmptl.dropout_p, training=training)
    return x, prep


def add_navi_latent(m, x: Tensor, z: Tensor, z_invalid: Tensor, training: bool) -> Tensor:
    zi = z_invalid.unsqueeze(-1)
    zz = mlp(m.mlp_in, z, training).masked_fill(zi, 0.0)
    h = mlp(m.mlp, torch.cat([x, zz], -1), training).masked_fill(zi, 0.0)
    return h + x


def _agent_policy_engine(model, hv, hp, hm, ag_attr6, ag_type, z, z_valid, dest, navi_valid, tl_tokens, mp, tl_pre, training) -> Tensor:
    """The agents' half of a policy step on the inference engine's schedule (TrafficBots.agent_policy: window PointNet chain,
    3 K-nearest searches, 4 layer chains + 8 attention launches, 2 heads chains) - for the stepping pass of training: no
    autograd, light tokens given (tl_pre), keyed dropouts emitted as stages / in-kernel in the torch path's site order."""
    from . import engine

    n, A, _ = hv.shape
    tl_feat, ids = tl_pre[0], tl_pre[1]
    u8 = torch.uint8
    out = dict(action_mean=torch.empty(n * A, 2, dtype=torch.float32, device=hp.device))
    # (tl_pre[2]: this step's K/V tables, made for all steps of the piece in one launch - the same row-local stages)
    tl_kv = tl_pre[2] if len(tl_pre) > 2 and tl_pre[2] is not None else engine.kv_tables(tl_feat.contiguous(), model.ag_encoder.tl_kv_layers())
    ctx = None
    if training and ST._DROP is not None:
        ctx = dict(seed=ST._DROP["seed"], site=ids[0], call=ids[1], step=ST._DROP["t0"])
    # per-rollout constants in the form the engine takes them: converted once per pass, not in each of its 90 steps (the pass's
    # map dict lives as long as the pass: ~8 small launches per step less)
    sc = mp.setdefault("_step_consts", {})

    def const(name, t, fn):
        k = (name, t.data_ptr(), tuple(t.shape))
        if k not in sc:
            sc[k] = fn(t)
        return sc[k]

    type_idx = const("type", ag_type, lambda t: t.to(u8).argmax(-1).to(u8).contiguous())
    zz = const("z", z, lambda t: t.reshape(n * A, -1).float().contiguous())
    zi = const("zi", z_valid, lambda t: (~t).reshape(-1).to(u8).contiguous())
    dd = const("dest", dest, lambda t: t.contiguous())

    def dest_feature(_):  # mlp_mp(map feature of the destination): no dropout in it, fixed over the rollout (TrafficBots.rollout_constants)
        M = mp["mp_token_pose"].shape[1]
        rows_ = (torch.arange(n, device=dd.device).unsqueeze(1) * M + dd).reshape(-1).to(torch.int32).contiguous()
        dst = torch.empty(n * A, model.hidden_dim, dtype=torch.float32, device=dd.device)
        ch = hip.Chain(16, 4 * model.hidden_dim + 4)
        model.navi_encoder.emit_dest_feature(ch, mp["mp_token_feature"].reshape(-1, model.hidden_dim), rows_, dst)
        ch.run(n * A)
        return dst

    rc = dict(dest_feature=const("destf", dd, dest_feature)) if HEADS_TILE else None
    engine.DROP_CTX = ctx
    try:
        # (the stepping pass keeps the VALU ring kernel in both classes: at its 1024 source rows - one row per wavefront, one wavefront
        # per SIMD - the matrix-core form measured 20.0 us per launch against 17.1, profiles/r05b_train_replay_timeline_bf16.txt; the
        # two passes then differ by the bf16 operand rounding of the batched forward, ~1e-2 of an action, far below the dropout noise)
        nv8 = navi_valid.view(u8) if navi_valid.dtype == torch.bool else navi_valid.to(u8)
        model.agent_policy(hv, hp, hm, ag_attr6, type_idx, zz, zi, dd, nv8.contiguous(), tl_tokens, mp, tl_kv, out, rollout_consts=rc)
    finally:
        engine.DROP_CTX = None
    if ctx is not None:
        ST._DROP["site"], ST._DROP["call"] = ctx["site"], ctx["call"]
    return out["action_mean"].view(n, A, 2)


def policy_step(model, hist, ag_attr6, ag_type, ag_valid, ag_pose, z, z_valid, dest, navi_valid, tl_tokens, mp, training: bool,
                T: int = 1, tl_pre=None, want_logits: bool = True):
    """traffic_bots.py:188-221 -> (action mean [n,A,2], tl logits [n,L,5]). T > 1: time-batched call - every per-entry
    argument has n = scenes x T entries ([scene][step] order), `mp` stays per scene and `tl_tokens` is expand_tl_tokens'.
    tl_pre = (tl_feat [n*L,128], (site, call) dropout ids after the light encoder): the light tokens were encoded ahead.
    want_logits=False (with tl_pre, autograd off): only the action means are needed - the agents' half runs on the engine's
    chains (`_agent_policy_engine`) and None is returned for the logits."""
    hv, hp, hm, ht = hist
    n, A, W = hv.shape
    d = model.hidden_dim
    L = ht.shape[1]
    assert tl_tokens.get("time_batch", 1) == T
    if tl_pre is not None and T == 1 and not want_logits and _chains_ok(hp):
        return _agent_policy_engine(model, hv, hp, hm, ag_attr6, ag_type, z, z_valid, dest, navi_valid, tl_tokens, mp, tl_pre, training), None
    if tl_pre is None:
        tl_feat = tl_encoder(model.tl_encoder, ht, tl_tokens, training)
    else:
        tl_feat = tl_pre[0].reshape(-1, d)
        if ST._DROP is not None:  # the agents' dropout sites keep the ids they have when the light encoder runs in place
            ST._DROP["site"], ST._DROP["call"] = tl_pre[1]
    feat, _ = agent_encoder(model.ag_encoder, hv, hp, hm, ag_attr6, mp, tl_tokens["tl_token_invalid_u8"], tl_tokens["tl_token_pose"],
                            tl_feat, training, T)
    # NaviEncoder (navigation.py:65-79): detached map feature of the destination + pose embedding of its relative pose
    ne = model.navi_encoder
    bi = (torch.arange(n, device=feat.device) // T).unsqueeze(1)
    mpf = mp["mp_token_feature"].detach() if ne.dest_detach_mp_feature else mp["mp_token_feature"]
    gp = mp["mp_token_pose"][bi, dest]
    c, s = torch.cos(ag_pose[..., 2]), torch.sin(ag_pose[..., 2])
    dx, dy = gp[..., 0] - ag_pose[..., 0], gp[..., 1] - ag_pose[..., 1]
    rel = torch.stack([dx * c + dy * s, dx * (-s) + dy * c, gp[..., 2] - ag_pose[..., 2]], -1).reshape(-1, 3).contiguous()
    pe = hip.pose_embed(rel, ne.pose_emb.pe_xy.freqs, ne.pose_emb.pe_yaw.freqs, ne.pose_emb.out_dim)
    navi = mlp(ne.mlp_mp, mpf[bi, dest].reshape(n * A, d)) + mlp(ne.mlp_pe, pe)
    feat = add_navi_latent(model.add_navi, feat, navi, ~navi_valid.reshape(-1), training)
    feat = add_navi_latent(model.add_latent, feat, z.reshape(n * A, -1), ~z_valid.reshape(-1), training)
    mask_type = ~(ag_type & ag_valid.unsqueeze(-1)).reshape(n * A, 3)
    mean = 0
    for i, m in enumerate(model.action_head.mlp_mean):
        mean = mean + mlp(m, feat).masked_fill(mask_type[:, i:i + 1], 0.0)
    sp = model.tl_state_predictor
    xt = tl_feat.detach() if sp.detach_tl_feature else tl_feat
    logits = torch.clamp(mlp(sp.mlp, xt).masked_fill(tl_tokens["tl_token_invalid"].reshape(-1, 1), 0.0), -3, 3)
    return mean.view(n, A, 2), logits.view(n, L, -1)


def latent_posterior(le, b, mp, tl_tokens, training: bool) -> DiagGaussian:
    r = le.temporal_down_sample_rate
    v, m, p, s = b["gt/ag_valid"][:, :, ::r], b["gt/ag_motion"][:, :, ::r], b["gt/ag_pose"][:, :, ::r], b["gt/tl_state"][:, :, ::r]
    te, ae = le.tl_encoder_post, le.ag_encoder_post
    n, A, _ = v.shape
    tl_feat = tl_encoder(te, te.states_to_hist(s, te.temp_window_size), tl_tokens, training)
    hv, hp, hm = ae.pad_hist(v, p, m, ae.temp_window_size)
    feat, _ = agent_encoder(ae, hv, hp, hm, b["sc/ag_attr"].float().contiguous(), mp, tl_tokens["tl_token_invalid_u8"],
                            tl_tokens["tl_token_pose"], tl_feat, training)
    valid = b["gt/ag_valid"].any(-1)
    mean = mlp(le.latent_dist_post.mlp_mean, feat).view(n, A, -1).masked_fill(~valid.unsqueeze(-1), 0.0)
    return DiagGaussian(mean, le.latent_dist_post.log_std, valid=valid)


def navi_predictor(npd, b, mp, training: bool) -> DestCategorical:
    """navigation.py:175-278 (dest)."""
    ag_valid, ag_pose, ag_motion = b["sc/ag_valid"], b["sc/ag_pose"].detach(), b["sc/ag_motion"].detach()
    n, A, W = ag_valid.shape
    assert W <= npd.temp_window_size
    dev, d = ag_pose.device, npd.hidden_dim
    from .models.agent_encoder import AgentEncoder

    hv, hp, hm = AgentEncoder.pad_hist(ag_valid, ag_pose, ag_motion, npd.temp_window_size)
    Wn = npd.temp_window_size
    f32, u8 = torch.float32, torch.uint8
    prep = dict(tok_pose=torch.empty(n, A, 3, dtype=f32, device=dev), tok_invalid=torch.empty(n, A, dtype=u8, device=dev),
                attr=torch.empty(n * A * Wn, 32, dtype=f32, device=dev), pe=torch.empty(n * A * Wn, npd.pose_emb.out_dim, dtype=f32, device=dev),
                row_invalid=torch.empty(n * A * Wn, dtype=u8, device=dev))
    hip.agent_prep(hv, hp, hm, b["sc/ag_attr"].float().contiguous(), None, npd.pose_emb.pe_xy.freqs, npd.pose_emb.pe_yaw.freqs,
                   npd.pose_emb.out_dim, prep)
    x = torch.cat([mlp(npd.input_encoder.mlp, prep["attr"][:, :npd.input_encoder.mlp.input_dim], training), prep["pe"]], -1)
    feat = pointnet(npd.temp_encoder, x.view(n * A, Wn, d), prep["row_invalid"].view(n * A, Wn).bool(), training)
    mpf = mp["mp_token_feature"].detach() if npd.detach_input else mp["mp_token_feature"]
    M = mpf.shape[1]
    tp, mpp = prep["tok_pose"], mp["mp_token_pose"]
    c, s = torch.cos(tp[..., 2])[:, :, None], torch.sin(tp[..., 2])[:, :, None]
    dx, dy = mpp[:, None, :, 0] - tp[:, :, None, 0], mpp[:, None, :, 1] - tp[:, :, None, 1]
    rel = torch.stack([dx * c + dy * s, dx * (-s) + dy * c, mpp[:, None, :, 2] - tp[:, :, None, 2]], -1).contiguous()  # [n, A, M, 3]
    # first Linear split into per-agent + per-polyline + per-pair terms: no [n, A, M, 384] tensor
    (lin1, ln1, act1), rest = npd.mlp.linear_layers()[0], npd.mlp.linear_layers()[1:]
    w1 = lin1.weight
    pa = linear(feat.view(n, A, d), w1[:, :d], None)
    pm = linear(mpf, w1[:, d:2 * d], lin1.bias)
    relu_in = bool(act1) and ln1 is None  # the first layer's relu in the launch that adds the broadcast terms
    x = NaviPairFirstLayer.apply(rel, w1[:, 2 * d:], pa, pm, npd.pose_rpe.pe_xy.freqs, npd.pose_rpe.pe_yaw.freqs, relu_in)
    if ln1 is not None:
        x = layer_norm(x, ln1)
    if act1 and not relu_in:
        x = F.relu(x)
    x = _drop(x, npd.mlp.dropout_p, training)
    for lin, lnm, act in rest:
        if act and lnm is None and hip.glue_ok(x) and lin.weight.shape[0] % 4 == 0:  # (as train_ops.mlp: LINEAR + relu + dropout as one launch)
            x = linear_relu_drop(x, lin.weight, lin.bias, npd.mlp.dropout_p, training)
            continue
        x = linear(x, lin.weight, lin.bias)
        if lnm is not None:
            x = layer_norm(x, lnm)
        if act:
            x = F.relu(x)
        x = _drop(x, npd.mlp.dropout_p, training)
    logits = x.squeeze(-1)
    ty, ag_type = mp["mp_token_type"], b["ref/ag_type"]
    tok_valid = ag_valid.any(-1)
    mp_mask = mp["mp_token_invalid"] | ~(ty[:, :, :5].any(-1))
    bad = (mp_mask[:, None] | (ag_type[:, :, 0:1] & ty[:, :, 3][:, None]) | (ag_type[:, :, 1:2] & ty[:, :, :4].any(-1)[:, None])
           | (ag_type[:, :, 2:3] & ty[:, :, :3].any(-1)[:, None]))
    logits = logits.masked_fill(bad, float("-inf")).masked_fill((~tok_valid).unsqueeze(-1) | bad.all(-1, keepdim=True), 0)
    return DestCategorical(logits=logits, valid=tok_valid)


# ------------------------------------------------------------------------------------------------ rollout + loss
def _bits(one_hot: Tensor) -> Tensor:
    w = (1 << torch.arange(one_hot.shape[-1], device=one_hot.device, dtype=torch.int32))
    return (one_hot.to(torch.int32) * w).sum(-1).to(torch.uint8)


def training_rollout(wm, b, mp, tl_tokens, z, z_valid, tf_mask: Tensor, step_end: int, policy=None, record: Optional[dict] = None,
                     need_hist: bool = True) -> Dict[str, Tensor]:
    """Closed-loop training rollout (waymo_motion.py:206-311 with training=True: model inputs detached, the only
    cross-step gradient path is the dynamics chain). The per-step state machine is elementwise torch on [n,A] tensors
    (it needs autograd); rule feedback = outside-map + dest-reached as in tbx_sim_step.
    policy(step, hist, valid, pose, navi_valid) -> (mean [n,A,2], logits [n,L,5]); None = policy_step with autograd, one step
    at a time (the reference's order of evaluation). record: dict of lists that receives every step's policy inputs.
    need_hist=False: the policy does not read the windows (they are not built)."""
    model, dyn, rc = wm.model, wm.dynamics, wm.hp.differentiable_reward
    gt_valid, gt_pose, gt_motion, tl_gt = b["gt/ag_valid"], b["gt/ag_pose"], b["gt/ag_motion"], b["gt/tl_state"]
    ag_type, ag_attr6, dest = b["ref/ag_type"], b["sc/ag_attr"].float().contiguous(), b["gt/ag_navi"]
    n, A, Tg = gt_valid.shape
    L, Tt = tl_gt.shape[1], tl_gt.shape[2]
    W, dev = model.temp_window_size, gt_pose.device
    dt = dyn.dt
    if getattr(dyn, "_max_act", None) is None or dyn._max_act.device != dev:  # host -> device once (not inside a captured step)
        dyn._max_act = torch.tensor([[a, y] for a, y in zip(dyn.max_acc, dyn.max_yaw_rate)], device=dev)
    max_act = dyn._max_act
    lim = (ag_type.unsqueeze(-1) * max_act).sum(2)
    bi = torch.arange(n, device=dev).unsqueeze(1)
    d_type = b["map/type"][bi, dest]
    d_dir = b["map/dir"][bi, dest][..., :2]
    d_dir = d_dir / torch.norm(d_dir, dim=-1, keepdim=True)
    d_pos, d_inv = b["map/pos"][bi, dest][..., :2], ~b["map/valid"][bi, dest]
    d_thresh = 50.0 * (1 - d_type[:, :, 4].float() * 0.8)
    bnd = b["map/boundary"]
    valid, disabled = gt_valid[:, :, 0], torch.zeros_like(gt_valid[:, :, 0])
    pose, motion = gt_pose[:, :, 0], gt_motion[:, :, 0]
    tl_bits = _bits(tl_gt)
    navi_valid = gt_valid.any(-1)
    outside, reached = torch.zeros_like(valid), torch.zeros_like(valid)
    hv = torch.zeros(n, A, W, dtype=torch.uint8, device=dev)
    hp, hm = torch.zeros(n, A, W, 3, device=dev), torch.zeros(n, A, W, 3, device=dev)
    ht = torch.full((n, L, W), 0xFF, dtype=torch.uint8, device=dev)
    tl_cur = tl_bits[:, :, 0]
    out = {k: [] for k in ("pred_valid", "pred_pose", "pred_motion", "tl_nll", "tl_nll_invalid", "reward", "reward_valid", "tf")}
    if policy is None:
        def policy(step, hist, valid_, pose_, navi_valid_):
            with _DropScope(n, 1, step, restart=_POLICY_SITE0):
                return policy_step(model, hist, ag_attr6, ag_type, valid_, pose_, z, z_valid, dest, navi_valid_, tl_tokens, mp,
                                   model.training)
    for step in range(1, step_end + 1):
        hist = None
        if need_hist:
            hv = torch.cat([hv[:, :, 1:], valid.to(torch.uint8).unsqueeze(2)], 2)
            hp = torch.cat([hp[:, :, 1:], pose.detach().unsqueeze(2)], 2)
            hm = torch.cat([hm[:, :, 1:], motion.detach().unsqueeze(2)], 2)
            ht = torch.cat([ht[:, :, 1:], tl_cur.unsqueeze(2)], 2)
            hist = (hv.contiguous(), hp.contiguous(), hm.contiguous(), ht.contiguous())
        if record is not None:
            for k, v in (("hv", hist[0]), ("hp", hist[1]), ("hm", hist[2]), ("ht", hist[3]), ("valid", valid), ("pose", pose.detach()),
                         ("navi_valid", navi_valid)):
                record.setdefault(k, []).append(v)
        mean, logits = policy(step, hist, valid, pose.detach(), navi_valid)
        inv1 = ~valid.unsqueeze(-1)
        action = (torch.tanh(mean) * lim).masked_fill(inv1, 0)
        acc, yr = action[..., 0], action[..., 1]
        v_t, th_t = motion[..., 0] + 0.5 * dt * acc, pose[..., 2] + 0.5 * dt * yr
        pose = (pose + dt * torch.stack([v_t * torch.cos(th_t), v_t * torch.sin(th_t), yr], -1)).masked_fill(inv1, 0)
        motion = torch.stack([motion[..., 0] + dt * acc, acc, yr], -1).masked_fill(inv1, 0)
        pred_valid, pred_pose, pred_motion = valid, pose, motion
        if step < Tg:
            ov = tf_mask[:, :, step] & ~disabled
            valid = valid | ov
            pose = pose.masked_fill(ov.unsqueeze(-1), 0) + gt_pose[:, :, step].masked_fill(~ov.unsqueeze(-1), 0)
            motion = motion.masked_fill(ov.unsqueeze(-1), 0) + gt_motion[:, :, step].masked_fill(~ov.unsqueeze(-1), 0)
            ov_log = tf_mask[:, :, step]
        else:
            ov_log = torch.zeros_like(valid)
        with torch.no_grad():
            if need_hist:
                tl_cur = tl_bits[:, :, step] if step < Tt else (1 << logits.argmax(-1)).to(torch.uint8)
            x, y = pred_pose[..., 0], pred_pose[..., 1]
            out_now = ((x > bnd[:, 1:2]) | (x < bnd[:, 0:1]) | (y > bnd[:, 3:4]) | (y < bnd[:, 2:3])) & pred_valid
            outside = outside | out_now
            dd = torch.norm(pred_pose[:, :, None, :2] - d_pos, dim=-1).masked_fill(d_inv, float("inf"))
            pos_ok = (dd < d_thresh.unsqueeze(-1)).any(-1)
            hf = torch.stack([torch.cos(pred_pose[..., 2]), torch.sin(pred_pose[..., 2])], -1)
            rot_ok = ((hf.unsqueeze(2) * d_dir).sum(-1).masked_fill(d_inv, 0) > 0.8660254037844387).any(-1)
            reach_now = (~reached) & pred_valid & ((d_type[:, :, :4].any(-1) & pos_ok & rot_ok) | (d_type[:, :, 4] & pos_ok))
            reached = reached | reach_now
        if step < Tg:  # rewards.py:58-74
            g_valid, g_pose, g_motion = gt_valid[:, :, step], gt_pose[:, :, step], gt_motion[:, :, step]
            r_valid = pred_valid & g_valid
            e_pos = F.smooth_l1_loss(g_pose[..., :2], pred_pose[..., :2], reduction="none").sum(-1)
            e_rot = 0.5 * (1 - torch.cos(g_pose[..., 2] - pred_pose[..., 2]))
            e_spd = F.smooth_l1_loss(g_motion[..., 0], pred_motion[..., 0], reduction="none")
            rew = ((-rc.l_pos.weight * e_pos).masked_fill(~r_valid, 0) + (-rc.l_rot.weight * e_rot).masked_fill(~r_valid, 0)
                   + (-rc.l_spd.weight * e_spd).masked_fill(~r_valid, 0))
            dis = out_now & ~g_valid
        else:
            r_valid, rew, dis = pred_valid, torch.zeros_like(pred_pose[..., 0]), out_now
        if step < Tt:
            nll = -Categorical(logits=logits, validate_args=False).log_prob(tl_gt[:, :, step].max(-1)[1])
            nll_inv = tl_tokens["tl_token_invalid"]
        else:
            nll, nll_inv = torch.zeros_like(logits[..., 0]), torch.ones_like(tl_tokens["tl_token_invalid"])
        for k, v in (("pred_valid", pred_valid), ("pred_pose", pred_pose), ("pred_motion", pred_motion), ("tl_nll", nll),
                     ("tl_nll_invalid", nll_inv), ("reward", rew), ("reward_valid", r_valid), ("tf", ov_log)):
            out[k].append(v)
        disabled = disabled | dis
        valid = valid & ~dis
        navi_valid = navi_valid & ~reach_now
    return {k: torch.stack(v, 2) for k, v in out.items()}


def tl_nll_all_steps(logits: Tensor, tl_gt: Tensor, tl_invalid: Tensor):
    """The light-state NLL of every step at once (waymo_motion.py:277-283): logits [n,T,L,5] of steps 1..T, tl_gt [n,L,Tt,5] one-hot,
    -> (nll [n,L,T], invalid [n,L,T]); steps without ground truth count as invalid."""
    n, T, L, _ = logits.shape
    S = min(T, tl_gt.shape[2] - 1)
    nll = torch.zeros(n, T, L, device=logits.device)
    inv = torch.ones(n, T, L, dtype=torch.bool, device=logits.device)
    if S > 0:
        idx = tl_gt[:, :, 1:S + 1].max(-1)[1].permute(0, 2, 1)  # [n,S,L]
        nll[:, :S] = -torch.log_softmax(logits[:, :S], -1).gather(-1, idx.unsqueeze(-1)).squeeze(-1)
        inv[:, :S] = tl_invalid.unsqueeze(1)
    return nll.permute(0, 2, 1), inv.permute(0, 2, 1)


# Pieces of the ahead-of-time light encoder (1: all steps at once on the one stream). Measured (bench.py --mode train): 1 -> 68.8,
# 2 -> 68.7, 3 -> 67.1, 5 -> 65.5 scenes/s - the light encoder's full-chip launches on the side stream do not fill the CUs the
# stepping pass leaves idle, they queue in front of its small launches (no preemption), so the overlap loses. Kept as an opt-in.
TL_CHUNKS = int(os.environ.get("TBX_TL_CHUNKS", "1"))
_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


KV_TABLES_TALL = os.environ.get("TBX_KV_TABLES_TALL", "1") != "0"  # (0: the exact-fp32 row chain, for A/B runs)


def _kv_tables_tall(x: Tensor, norms_and_attns) -> Tensor:
    """engine.kv_tables (out[:, l * 256 ..] = LN_l(x) W_kv,l^T + b_kv,l: transformer_rpe.py:220-223 + attention_rpe.py:92-98) for the
    10^5 light-token rows of a whole piece of the stepping pass: per layer tbx_layernorm_fwd + tbx_tall_linear (the three-product
    split-bf16 matrix path: the arithmetic of the inference schedule's tbx_tl_tail_tile, < 3e-5 of sum |x||w|) straight into the layer's
    column block - byte-bound launches instead of ONE exact-fp32 row chain at the fp32 MFMA rate (184,320 rows: 1.2 ms -> ~0.6)."""
    from . import engine

    rows, L = x.shape[0], len(norms_and_attns)
    if not (KV_TABLES_TALL and TALL_LINEAR and rows >= WGRAD_MIN_ROWS and engine.kv_dtype() == torch.float32 and hip.tall_linear_ok(x, D, 2 * D)
            and hip.layernorm_bwd_ok(x)):
        return engine.kv_tables(x, norms_and_attns)
    out = torch.empty(rows, 2 * D * L, dtype=torch.float32, device=x.device)
    for l, (nm, attn) in enumerate(norms_and_attns):
        y = hip.layernorm_fwd(x, nm.weight, nm.bias, nm.eps)[0]
        hip.tall_linear(y, attn.in_proj_weight[D:], attn.in_proj_bias[D:], out=out[:, l * 2 * D:(l + 1) * 2 * D])
    return out


def training_rollout_batched(wm, b, mp, tl_tokens, z, z_valid, tf_mask: Tensor, step_end: int) -> Dict[str, Tensor]:
    """The same rollout, time-batched (module docstring): a step-by-step pass without autograd that records every step's
    policy inputs, then the T policy evaluations of every scene as one differentiated batch of n x T entries in [scene][step]
    order, then the per-step dynamics / reward chain on the batched means (the only cross-step gradient path)."""
    model = wm.model
    T = step_end
    n = b["gt/ag_valid"].shape[0]
    flat = lambda x: x.reshape(n * T, *x.shape[2:]).contiguous()
    rep = lambda x: x.repeat_interleave(T, 0)
    ag_attr6, ag_type, dest = b["sc/ag_attr"].float().contiguous(), b["ref/ag_type"], b["gt/ag_navi"]
    tl_T = expand_tl_tokens(model.tl_encoder, tl_tokens, mp, T)
    # The lights are teacher-forced while ground truth lasts (tl_cur = gt state for step < Tt), so with step_end <= Tt every
    # light window is known before the rollout: the light encoder of all T steps runs ONCE, differentiated, ahead of pass 1,
    # which then only steps the agents' half of the policy on its (detached) tokens.
    tl_gt = b["gt/tl_state"]
    L, Tt, W = tl_gt.shape[1], tl_gt.shape[2], model.temp_window_size
    tl_pre, tl_steps, ht_all = None, None, None
    if step_end <= Tt:
        pad = torch.full((n, L, W), 0xFF, dtype=torch.uint8, device=tl_gt.device)
        ht_all = torch.cat([pad, _bits(tl_gt)[:, :, :T]], 2).unfold(2, W, 1)[:, :, 1:T + 1]  # [n, L, T, W]: window of step s = states s-W .. s-1
        ht_all = ht_all.permute(0, 2, 1, 3).reshape(n * T, L, W).contiguous()
    tl_chunks = None  # [(first step index, [n, Tc, L, d] detached features, event or None)]
    if ht_all is not None and getattr(wm, "tl_encoder_ahead", True):
        # ... in TL_CHUNKS pieces along time: the first on this stream, the others on a side stream WHILE the stepping pass below runs
        # (its launches are latency-bound on a quarter of the CUs; the light encoder's are the big-batch kind). The stepping pass waits
        # for a piece's event before its first step. Same keyed masks (they are keyed by the absolute step), same per-row arithmetic.
        n_chunk = TL_CHUNKS if (ht_all.is_cuda and T >= 3 * TL_CHUNKS) else 1
        bounds = [round(i * T / n_chunk) for i in range(n_chunk + 1)]
        main, side = (torch.cuda.current_stream(), _side_stream(ht_all.device)) if n_chunk > 1 else (None, None)
        ht4, shared_kv, feats, tl_chunks, ids = ht_all.view(n, T, L, W), {}, [], [], None
        for c in range(n_chunk):
            c0, Tc = bounds[c], bounds[c + 1] - bounds[c]
            if c == 1:
                side.wait_stream(main)  # fork: the map K/V tables of the lights' cross attention were made (and cached) by piece 0
            with torch.cuda.stream(side) if c > 0 else contextlib.nullcontext():
                tl_Tc = tl_T
                if n_chunk > 1:
                    tl_Tc = expand_tl_tokens(model.tl_encoder, tl_tokens, mp, Tc)
                    tl_Tc["_kv_cache"] = shared_kv
                with _DropScope(n * Tc, Tc, 1 + c0, restart=_POLICY_SITE0):
                    f = tl_encoder(model.tl_encoder, ht4[:, c0:c0 + Tc].reshape(n * Tc, L, W).contiguous(), tl_Tc, model.training)
                    ids = (ST._DROP["site"], ST._DROP["call"]) if ST._DROP is not None else None
                ev = None
                if c > 0:
                    ev = torch.cuda.Event()
                    ev.record()
            feats.append(f.view(n, Tc, L, -1))
            tl_chunks.append((c0, feats[-1].detach(), ev))
    fused = getattr(wm, "fused_train_chain", True) and ht_all is not None  # (lights from their own logits: the torch state machine)
    chain = TrainChain(wm, b, tf_mask, T) if fused else None
    rec: Dict[str, List[Tensor]] = {}
    with torch.no_grad():  # pass 1: own K/V caches (its graph-less tables must not reach the differentiated pass)
        mp1, tl1 = dict(mp, _kv_cache={}), dict(tl_tokens, _kv_cache={})

        kv_pieces: Dict[int, Tensor] = {}

        def tl_of(step):  # the light tokens of `step` from the piece that holds it (joined on first use) + their K/V tables
            for i in range(len(tl_chunks) - 1, -1, -1):
                c0, f, ev = tl_chunks[i]
                if step - 1 >= c0:
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                        tl_chunks[i] = (c0, f, None)
                    if i not in kv_pieces and ST.NOGRAD_CHAINS and f.is_cuda and not torch.is_grad_enabled():
                        # the K/V rows the agents' layers read, for ALL steps of the piece in one launch (time-major: a step's
                        # tables are then a contiguous [n * L, 1024] block) instead of one launch per step (90 x 37 us)
                        from . import engine

                        ft = f.permute(1, 0, 2, 3).contiguous()  # [Tc, n, L, d]
                        kv_pieces[i] = _kv_tables_tall(ft.view(-1, ft.shape[-1]), model.ag_encoder.tl_kv_layers()).view(ft.shape[0], n * L, -1)
                    kv = kv_pieces[i][step - 1 - c0] if i in kv_pieces else None
                    # (with the tables at hand the engine path does not read the features: no per-step gather of them)
                    return (f[:, step - 1 - c0].reshape(n * L, -1) if kv is None else f[:, step - 1 - c0]), kv

        def policy1(step, hist, valid_, pose_, navi_valid_):
            pre = None
            if tl_chunks is not None:
                tf_, kv_ = tl_of(step)
                pre = (tf_, ids, kv_)
            with _DropScope(n, 1, step, restart=_POLICY_SITE0):
                return policy_step(model, hist, ag_attr6, ag_type, valid_, pose_, z.detach(), z_valid, dest, navi_valid_, tl1, mp1,
                                   model.training, tl_pre=pre, want_logits=not fused)

        if fused:  # the state machine is tbx_train_chain: per step the policy, then ONE launch
            ht_steps = ht_all.view(n, T, L, W)
            for step in range(1, T + 1):
                hv, hp, hm, valid_, pose_, navi_valid_ = chain.before(step)
                # (the light windows are only read when the lights are encoded in the step: not with their tokens made ahead)
                mean1, _ = policy1(step, (hv, hp, hm, ht_steps[:, step - 1] if tl_chunks is not None else ht_steps[:, step - 1].contiguous()),
                                   valid_, pose_, navi_valid_)
                chain.step(step, mean1)
            inputs = chain.windows()
        else:
            training_rollout(wm, b, mp1, tl1, z.detach(), z_valid, tf_mask, step_end, policy=policy1, record=rec)
            st = {k: torch.stack(v, 1) for k, v in rec.items()}  # [n, T, ...]
            inputs = (flat(st["hv"]), flat(st["hp"]), flat(st["hm"]), flat(st["valid"]), flat(st["pose"]), flat(st["navi_valid"]))
    if tl_chunks is not None:
        if len(tl_chunks) > 1:
            torch.cuda.current_stream().wait_stream(_side_stream(ht_all.device))  # join
        tl_feat_all = feats[0] if len(feats) == 1 else torch.cat(feats, 1)
        tl_pre = (tl_feat_all.reshape(n * T * L, -1), ids)
    hv, hp, hm, valid_all, pose_all, navi_all = inputs
    ht_in = ht_all if ht_all is not None else flat(st["ht"])
    with _DropScope(n * T, T, 1, restart=_POLICY_SITE0):
        mean, logits = policy_step(model, (hv, hp, hm, ht_in), rep(ag_attr6).contiguous(), rep(ag_type), valid_all, pose_all, rep(z),
                                   rep(z_valid), rep(dest), navi_all, tl_T, mp, model.training, T=T, tl_pre=tl_pre)
    mean, logits = mean.view(n, T, *mean.shape[1:]), logits.view(n, T, *logits.shape[1:])
    if fused:
        ro = chain.outputs(TrainChainFn.apply(mean, chain))
        ro["tl_nll"], ro["tl_nll_invalid"] = tl_nll_all_steps(logits, tl_gt, tl_tokens["tl_token_invalid"])
        return ro
    return training_rollout(wm, b, mp, tl_tokens, z, z_valid, tf_mask, step_end, need_hist=False,
                            policy=lambda step, hist, v, p, nv: (mean[:, step - 1], logits[:, step - 1]))


def training_loss(cfg, ro, navi_pred: DestCategorical, navi_gt, post: DiagGaussian, prior: DiagGaussian,
                  counts: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """metrics/training.py:74-189 + metrics/loss.py:39-77 (default weights / switches). counts: receives every term's normaliser
    (the number of entries its sum runs over, un-clamped) - a term of a batch is the count-weighted mean of its per-scene terms."""
    # A term whose counter is zero (a batch without a valid light / agent) is left out by the reference (training.py:166-186:
    # `if counter > 0`): sum = 0 over a count clamped to 1 gives that 0 without a host branch (the step stays capturable).
    cnt = lambda m: m.sum().clamp(min=1)
    lv = ro["pred_valid"].clone()
    lv[:, :, : cfg.step_training_start] &= False
    if not cfg.loss_for_teacher_forcing:
        lv &= ~ro["tf"]
    any_valid = lv.any(-1)
    P, Q = post.distribution, prior.distribution
    dP = Independent(Normal(P.base_dist.loc.detach(), P.base_dist.scale.detach(), validate_args=False), 1, validate_args=False)
    dQ = Independent(Normal(Q.base_dist.loc.detach(), Q.base_dist.scale.detach(), validate_args=False), 1, validate_args=False)
    e0 = torch.clamp(kl_divergence(dP, Q), min=cfg.kl_free_nats)
    e1 = torch.clamp(kl_divergence(P, dQ), min=cfg.kl_free_nats)
    kv = (post.valid if cfg.kl_for_unseen_agent else prior.valid) & any_valid
    vae_kl = cfg.w_vae_kl * (e0 + cfg.kl_balance_scale * e1).masked_fill(~kv, 0).sum() / cnt(kv)
    rv = lv & ro["reward_valid"]
    reward = cfg.w_diffbar_reward * ro["reward"].masked_fill(~rv, 0).sum() / cnt(rv)
    nv = navi_pred.valid & any_valid
    navi = cfg.w_navi * (-navi_pred.log_prob(navi_gt)).masked_fill(~nv, 0).sum() / cnt(nv)
    tv = ~ro["tl_nll_invalid"]
    tl = cfg.w_tl_state * ro["tl_nll"].masked_fill(~tv, 0).sum() / cnt(tv)
    if counts is not None:
        counts.update(vae_kl=kv.sum(), diffbar_reward=rv.sum(), navi_loss=nv.sum(), tl_state_loss=tv.sum())
    return {"loss": vae_kl - reward + navi + tl, "vae_kl": vae_kl, "diffbar_reward": reward, "navi_loss": navi, "tl_state_loss": tl}


def training_step(wm, raw_batch: Dict[str, Tensor], noise: Optional[Tensor] = None, use_prior: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """waymo_motion.py:313-385. Dropout: residual / FFN / MLP dropouts through torch, the attention-probability dropout inside
    the HIP attention kernels (seed on the device, one call id per attention call of the step). `noise` [n,A,latent] / `use_prior` (0-d bool tensor) are the two host-drawn random
    inputs of a step as device tensors: a captured step (pl_modules/data_parallel.GraphedTrainStep) refills them before
    every replay; left None they are drawn here from the CPU generator like the reference's CPU path does."""
    ST._FOLD_CACHE, ST._KV16 = {}, {}
    ST._PREC = getattr(wm, "train_precision", None)
    if ST._PREC not in (None, "bf16", "fp32"):
        raise ValueError(f"train_precision {ST._PREC!r}: 'bf16' (autocast-class contractions) or 'fp32'")
    if wm.model.training:
        seed = getattr(wm, "attn_dropout_seed", None)  # a captured step owns a static seed tensor and refills it per replay
        if seed is None:
            seed = torch.empty(1, dtype=torch.int64, device=next(wm.model.parameters()).device).random_()
        ST._DROP = {"seed": seed, "call": 0, "site": 0, "n_batch": 0, "tb": 1, "t0": 0}
    # weight images packed once per step (inside a captured step too): the Parameter-sourced ones of the previous step's list in ONE launch
    # now (hip_base.open_pack_scope), the folded attention weights' per module (train_ops.fold_attention_weights: hip_base.pack_group)
    scope = hip_base.open_pack_scope(wm.model, ("training_step", ST._PREC, bool(wm.model.training)))
    try:
        return _training_step(wm, raw_batch, noise, use_prior)
    finally:
        hip_base.close_pack_scope(scope)
        # the backward of this step (before any optimizer step) may go on with the same images: backward_pack_scope()
        wm._pack_scope = scope
        ST._FOLD_CACHE, ST._DROP, ST._PREC, ST._KV16 = None, None, None, None


class backward_pack_scope:
    """with backward_pack_scope(wm): loss.backward() - the backward's weight images (the W^T image of every tall LINEAR's input gradient)
    in the scope of the training_step that produced `loss`: packed with the forward's (their requests join its list), never taken from
    the per-Parameter cache - a captured backward must contain its packing (pl_modules/data_parallel.GraphedTrainStep._fwd_bwd)."""

    def __init__(self, wm):
        self.wm = wm

    def __enter__(self):
        scope = getattr(self.wm, "_pack_scope", None)
        self.wm._pack_scope = None  # (one backward per scope: the optimizer step that follows ends the images' validity)
        hip_base.PACK_SCOPE = scope if scope is not None else {}
        return hip_base.PACK_SCOPE

    def __exit__(self, *exc):
        hip_base.close_pack_scope()


def _training_step(wm, raw_batch, noise, use_prior) -> Dict[str, Tensor]:
    model, hp = wm.model, wm.hp
    tr = model.training
    if "sc/mp_valid" in raw_batch:  # already re-keyed (a captured step pre-processes eagerly: its index tensors come from the host)
        b = raw_batch
    else:
        with torch.no_grad():
            b = wm.pre_processing(raw_batch)
    if ST._DROP is not None:
        ST._DROP["n_batch"] = b["sc/mp_valid"].shape[0]
    mp = map_encoder(model.mp_encoder, b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"], tr)
    tl_tokens = tl_pre_compute(model.tl_encoder, b["gt/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], mp)
    mp["_kv_cache"], tl_tokens["_kv_cache"] = {}, {}  # map K/V tables: once per training step, shared by all 90 steps
    post = latent_posterior(model.latent_encoder, b, mp, tl_tokens, tr)
    pr = model.latent_encoder.latent_dist_prior
    valid_hist = b["sc/ag_valid"].any(-1)
    prior = DiagGaussian(pr.mean.expand(*valid_hist.shape, -1), pr.log_std, valid=valid_hist)
    # rsample with the noise drawn from the CPU generator, as the reference's CPU path does (same stream under the same
    # seed); one [n, A, 16] host-to-device copy per training step
    if use_prior is None:
        lat = prior if torch.rand(1) < hp.p_training_rollout_prior else post
        l_mean, l_std, l_valid = lat.mean, lat.stddev, lat.valid
    else:  # the same choice as a device-side select (no host branch inside a captured step)
        l_mean = torch.where(use_prior, prior.mean, post.mean)
        l_std = torch.where(use_prior, prior.stddev.expand_as(post.mean), post.stddev.expand_as(post.mean))
        l_valid = torch.where(use_prior, prior.valid, post.valid)
    if noise is None:
        noise = torch.randn(l_mean.shape).to(l_mean.device)
    z = l_mean + l_std * noise
    navi_pred = navi_predictor(model.navi_predictor, b, mp, tr)
    tf = wm.teacher_forcing_training
    tf.init(ag_valid=b["gt/ag_valid"], ag_pose=b["gt/ag_pose"], ag_motion=b["gt/ag_motion"], tl_state=b["gt/tl_state"],
            current_epoch=wm.current_epoch)
    rollout = training_rollout_batched if getattr(wm, "time_batched_training", True) else training_rollout
    ro = rollout(wm, b, mp, tl_tokens, z, l_valid, tf.ag_teacher_forcing, hp.time_step_end)
    wm.last_counts = {}
    return training_loss(hp.training_metrics, ro, navi_pred, b["gt/ag_navi"], post, prior, counts=wm.last_counts)
