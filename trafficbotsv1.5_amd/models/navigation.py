"""`NaviEncoder` / `NaviPredictor` (models/navigation.py:18-322), navi_mode=dest.
NaviEncoder is per step (destination token feature + relative-pose embedding); NaviPredictor runs once per scene
(one LN-MLP logit per (agent, polyline) pair) and is scheduled as a row chain over the A*M pair rows."""
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .. import hip
from ..engine import emit_mlp, emit_pointnet
from ..hip import BUF0, BUF1, Chain
from ..utils.pose_emb import PoseEmb
from .modules.distributions import DestCategorical
from .modules.input_encoder import InputEncoder
from .modules.mlp import MLP
from .modules.polyline_encoder import PolylineEncoder


class NaviEncoder(nn.Module):
    def __init__(self, hidden_dim, navi_mode: str, navi_dim: Optional[int], pairwise_relative: bool,
                 dest_detach_mp_feature: bool, mp_pose_emb: PoseEmb, pose_rpe: PoseEmb) -> None:
        super().__init__()
        if navi_mode != "dest" or not pairwise_relative:
            raise NotImplementedError("the MI355X path implements navi_mode=dest (pairwise relative)")
        self.navi_mode, self.pairwise_relative, self.dest_detach_mp_feature = navi_mode, pairwise_relative, dest_detach_mp_feature
        self.require_update, self.dummy, self.hidden_dim = True, False, hidden_dim
        self.mlp_mp = MLP([hidden_dim, hidden_dim], end_layer_activation=False)
        self.pose_emb = pose_rpe
        self.mlp_pe = MLP([pose_rpe.out_dim, hidden_dim], end_layer_activation=False)

    def emit_dest_feature(self, ch: Chain, mp_feat_flat: Tensor, navi_row: Tensor, out: Tensor):
        """mlp_mp(mp_feature[dest]) -> out [rows, d]: fixed while the destinations are (a whole rollout with
        pred_navi_after_reached off), so the engine evaluates it once and passes it to `emit(..., dest_feature=...)`."""
        d = self.hidden_dim
        l_mp = self.mlp_mp.linear_layers()[0][0]
        ch.load(mp_feat_flat, BUF0, 0, n=d, row_idx=navi_row)
        ch.linear(BUF0, 0, BUF0, d, l_mp.weight, l_mp.bias)
        ch.store(BUF0, d, d, out)

    def emit(self, ch: Chain, mp_feat_flat: Tensor, navi_row: Tensor, navi_pe: Tensor, dest_feature: Optional[Tensor] = None):
        """navi feature -> BUF0[:, d:2d]: mlp_mp(mp_feature[dest]) + mlp_pe(pe(rel pose of dest)). navigation.py:65-79."""
        d = self.hidden_dim
        l_mp, l_pe = self.mlp_mp.linear_layers()[0][0], self.mlp_pe.linear_layers()[0][0]
        if dest_feature is not None:
            ch.load2(navi_pe, BUF0, 0, dest_feature, BUF0, d)  # both inputs in one load stage (one memory round trip)
        else:
            ch.load(mp_feat_flat, BUF0, 0, n=d, row_idx=navi_row)
            ch.linear(BUF0, 0, BUF0, d, l_mp.weight, l_mp.bias)
            ch.load(navi_pe, BUF0, 0, n=d)
        ch.linear(BUF0, 0, BUF0, d, l_pe.weight, l_pe.bias, accum=True)

    def forward(self, ag_navi: Tensor, ag_pose: Tensor, mp_token_feature: Tensor, mp_token_pose: Tensor) -> Tensor:
        n, A = ag_navi.shape
        M, d, dev = mp_token_pose.shape[1], self.hidden_dim, ag_pose.device
        gp = mp_token_pose[torch.arange(n, device=dev).unsqueeze(1), ag_navi]  # [n,A,3] gather (host-side indexing only)
        # relative pose of the destination in the agent frame, through the KNN kernel's rel-pose path with K targets = 1
        # is overkill; the fused engine computes it in tbx_agent_prep. Stand-alone call: tiny torch index + embed kernel.
        c, s = torch.cos(ag_pose[..., 2]), torch.sin(ag_pose[..., 2])
        dx, dy = gp[..., 0] - ag_pose[..., 0], gp[..., 1] - ag_pose[..., 1]
        rel = torch.stack([dx * c + dy * s, dx * (-s) + dy * c, gp[..., 2] - ag_pose[..., 2]], -1).reshape(-1, 3).contiguous()
        pe = hip.pose_embed(rel.float(), self.pose_emb.pe_xy.freqs, self.pose_emb.pe_yaw.freqs, self.pose_emb.out_dim)
        rows = (torch.arange(n, device=dev).unsqueeze(1) * M + ag_navi).reshape(-1).to(torch.int32).contiguous()
        out = torch.empty(n * A, d, dtype=torch.float32, device=dev)
        ch = Chain(16, 2 * d + 4)
        self.emit(ch, mp_token_feature.reshape(-1, d).contiguous().float(), rows, pe)
        ch.store(BUF0, d, d, out)
        ch.run(n * A)
        return out.view(n, A, d)


class NaviPredictor(nn.Module):
    def __init__(self, navi_mode: str, detach_input: bool, rnn_res_add: bool, n_layer_tf: int, n_layer_mlp: int,
                 navi_dim: Optional[int], mlp_use_layernorm: bool, k_tgt_knn: float, k_dist_limit: float, ag_encoder,
                 goal_log_std: float, pose_rpe: PoseEmb) -> None:
        super().__init__()
        if navi_mode != "dest" or not ag_encoder["pairwise_relative"] or ag_encoder["temp_window_size"] <= 0:
            raise NotImplementedError("the MI355X path implements navi_mode=dest (HPTR, pairwise relative)")
        self.navi_mode, self.detach_input, self.pose_rpe = navi_mode, detach_input, pose_rpe
        self.temp_window_size = ag_encoder["temp_window_size"]
        self.hidden_dim = hidden_dim = ag_encoder["hidden_dim"]
        input_encoder, temp_encoder = ag_encoder["input_encoder"], ag_encoder["temp_encoder"]
        self.pose_emb = PoseEmb(pe_dim=hidden_dim // 2, **ag_encoder["pose_emb"])
        attr_dim = ag_encoder["ag_attr_dim"] + ag_encoder["ag_motion_dim"] + self.temp_window_size
        self.register_buffer("hist_ohe", torch.eye(self.temp_window_size))
        self.temp_encoder = PolylineEncoder(hidden_dim=hidden_dim, tf_cfg=ag_encoder["tf_cfg"], **temp_encoder)
        self.input_encoder = InputEncoder(hidden_dim=hidden_dim, attr_dim=attr_dim, pe_dim=self.pose_emb.out_dim, **input_encoder)
        self.mlp = MLP([2 * hidden_dim + pose_rpe.out_dim] + [hidden_dim] * (n_layer_mlp - 1) + [1], end_layer_activation=False,
                       use_layernorm=mlp_use_layernorm)

    def forward(self, ag_valid: Tensor, ag_attr: Tensor, ag_motion: Tensor, ag_pose: Tensor, mp_token_invalid: Tensor,
                mp_token_feature: Tensor, mp_token_pose: Tensor, ag_type: Tensor, mp_token_type: Tensor, **kwargs
                ) -> DestCategorical:
        """Destination logits [n, A, M] (navigation.py:175-278)."""
        from .agent_encoder import AgentEncoder

        n, A, n_step = ag_valid.shape
        M, d, dev, W = mp_token_pose.shape[1], self.hidden_dim, ag_pose.device, self.temp_window_size
        tok_valid = ag_valid.any(-1)
        # token pose = last valid pose of the FULL history; the window keeps the latest W steps (navigation.py:201-214)
        if n_step > W:
            idx_last = n_step - 1 - torch.max(ag_valid.flip(2).int(), dim=2)[1]
            ar = torch.arange
            full_tok = ag_pose[ar(n, device=dev)[:, None], ar(A, device=dev)[None, :], idx_last]
            full_tok = full_tok.masked_fill(~tok_valid.unsqueeze(-1), 0)
            assert torch.equal(ag_valid[:, :, -W:].any(-1), tok_valid), "token frame outside the window is not supported"
            ag_valid, ag_pose, ag_motion = ag_valid[:, :, -W:], ag_pose[:, :, -W:], ag_motion[:, :, -W:]
            del full_tok
        hv, hp, hm = AgentEncoder.pad_hist(ag_valid, ag_pose, ag_motion, W)
        f32, u8 = torch.float32, torch.uint8
        prep = dict(tok_pose=torch.empty(n, A, 3, dtype=f32, device=dev), tok_invalid=torch.empty(n, A, dtype=u8, device=dev),
                    attr=torch.empty(n * A * W, 32, dtype=f32, device=dev),
                    pe=torch.empty(n * A * W, self.pose_emb.out_dim, dtype=f32, device=dev),
                    row_invalid=torch.empty(n * A * W, dtype=u8, device=dev))
        hip.agent_prep(hv, hp, hm, ag_attr.float().contiguous(), None, self.pose_emb.pe_xy.freqs, self.pose_emb.pe_yaw.freqs,
                       self.pose_emb.out_dim, prep)
        feat = torch.empty(n * A, d, dtype=f32, device=dev)
        ch = Chain(hip.group_tile_rows(W, n * A), d + 4)
        cur = self.input_encoder.emit(ch, prep["attr"], prep["pe"])
        emit_pointnet(ch, self.temp_encoder, prep["row_invalid"], feat, x_buf=cur)
        ch.run(n * A * W, group_rows=W)
        # pair rows (a, m). The first Linear of the pair MLP acts on [agent feature | map feature | pose embedding of the map token
        # in the agent frame] (navigation.py:245-262): its weight splits into W_a | W_m | W_e, so the per-agent term W_a f_a
        # ([n*A, 128]) and the per-polyline term W_m f_m + b ([n*M, 128]) are computed once per token and only the per-pair
        # embedding term remains a per-pair product - a third of the first layer's multiplies, no 384-wide pair rows.
        mp_inv_u8 = mp_token_invalid.to(u8).contiguous()
        mp_pose = mp_token_pose.float().contiguous()
        lins = self.mlp.linear_layers()
        (lin1, ln1, act1), rest = lins[0], lins[1:]
        w1 = lin1.weight
        from ..engine import row_chain

        pa = torch.empty(n * A, d, dtype=f32, device=dev)
        ch = row_chain(n * A, d + 4, d + 4, d + 4)
        ch.load(feat, BUF0, 0, n=d)
        ch.linear(BUF0, 0, BUF1, 0, w1[:, :d])
        ch.store(BUF1, 0, d, pa)
        ch.run(n * A)
        mpf = mp_token_feature.reshape(-1, d).contiguous().float()
        pm = torch.empty(mpf.shape[0], d, dtype=f32, device=dev)
        ch = row_chain(mpf.shape[0], d + 4, d + 4, d + 4)
        ch.load(mpf, BUF0, 0, n=d)
        ch.linear(BUF0, 0, BUF1, 0, w1[:, d:2 * d], lin1.bias)
        ch.store(BUF1, 0, d, pm)
        ch.run(mpf.shape[0])
        logits = torch.empty(n * A * M, 1, dtype=f32, device=dev)
        rel = _all_rel_pose(prep["tok_pose"], mp_pose)  # [n*A*M, 3]
        emb = hip.pose_embed(rel, self.pose_rpe.pe_xy.freqs, self.pose_rpe.pe_yaw.freqs, self.pose_rpe.out_dim)
        ch = Chain(16, 2 * d + 4)
        ch.load(pa, BUF1, 0, n=d, row_div=M)
        ch.load(pm, BUF1, 0, n=d, accum=True, batch_mod=(A * M, M))
        ch.load(emb, BUF0, 0, n=d)
        ch.linear(BUF0, 0, BUF1, 0, w1[:, 2 * d:], accum=True)
        cur, other = BUF1, BUF0
        if ln1 is not None:
            ch.layernorm(cur, 0, cur, 0, ln1.weight, ln1.bias, ln1.eps)
        if act1:
            ch.clamp(cur, 0, d, 0.0, float("inf"))
        for lin, lnm, act in rest:
            no = lin.weight.shape[0]
            if lnm is None:
                ch.linear(cur, 0, other, 0, lin.weight, lin.bias, relu=act)
            else:
                ch.linear(cur, 0, other, 0, lin.weight, lin.bias)
                ch.layernorm(other, 0, other, 0, lnm.weight, lnm.bias, lnm.eps)
                if act:
                    ch.clamp(other, 0, no, 0.0, float("inf"))
            cur, other = other, cur
        ch.store(cur, 0, 1, logits)
        ch.run(n * A * M)
        logits = logits.view(n, A, M)
        ty = mp_token_type
        mp_mask = mp_token_invalid | ~(ty[:, :, :5].any(-1))
        bad = (mp_mask[:, None] | (ag_type[:, :, [0]] & ty[:, :, 3][:, None]) | (ag_type[:, :, [1]] & ty[:, :, :4].any(-1)[:, None])
               | (ag_type[:, :, [2]] & ty[:, :, :3].any(-1)[:, None]))
        logits = logits.masked_fill(bad, float("-inf"))
        logits = logits.masked_fill((~tok_valid).unsqueeze(-1) | bad.all(-1, keepdim=True), 0)
        return DestCategorical(logits=logits, valid=tok_valid)


def _all_rel_pose(tok_pose: Tensor, mp_pose: Tensor) -> Tensor:
    """Dense [n, A, M, 3] relative poses (utils/rpe.py:26-33) for the once-per-scene destination head."""
    c, s = torch.cos(tok_pose[..., 2])[:, :, None], torch.sin(tok_pose[..., 2])[:, :, None]
    dx = mp_pose[:, None, :, 0] - tok_pose[:, :, None, 0]
    dy = mp_pose[:, None, :, 1] - tok_pose[:, :, None, 1]
    yaw = mp_pose[:, None, :, 2] - tok_pose[:, :, None, 2]
    return torch.stack([dx * c + dy * s, dx * (-s) + dy * c, yaw], -1).reshape(-1, 3).contiguous()
