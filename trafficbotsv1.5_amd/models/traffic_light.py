"""`TrafficLightEncoder` / `TrafficLightStatePredictor` (models/traffic_light.py:15-287), tl_mode=lane, HPTR.
Static per scene (pre_compute): tl2tl / tl2mp KNN sets, their pose embeddings, and the K/V tables of the map tokens
for each tl2mp attention layer. Per step: state window -> PointNet token -> 4 dec_cross_attn layers."""
import os
from typing import Callable, Dict, Optional

import torch
from torch import Tensor, nn

from .. import hip
from ..engine import D, FIRST_PROJ_LDW, SelfKnn, emit_first_proj, emit_pointnet, first_proj_buffers, front_ok, front_proj_buffers, kv_dtype, kv_tables, run_block, tile_proj_part, tile_rows_ok, tile_small_ok
from ..hip import BUF0, BUF1, Chain, Seg
from ..utils.pose_emb import PoseEmb
from .modules.input_encoder import InputEncoder
from .modules.mlp import MLP
from .modules.polyline_encoder import PolylineEncoder
from .modules.transformer_rpe import TransformerBlockRPE


class TrafficLightEncoder(nn.Module):
    def __init__(self, hidden_dim: int, tl_state_dim: int, pairwise_relative: bool, tl_mode: str, pose_emb, input_encoder,
                 pose_rpe: Optional[PoseEmb], temp_encoder, temp_window_size: int, temp_stack_input: bool, tf_cfg,
                 n_tgt_knn: int, k_tgt_knn_tl2tl: float, k_tgt_knn_tl2mp: float, dist_limit: float, k_dist_limit: float,
                 n_layer_tf: int, tl_lane_detach_mp_feature: int) -> None:
        super().__init__()
        if tl_mode != "lane" or not pairwise_relative or temp_window_size <= 0 or temp_stack_input:
            raise NotImplementedError("the MI355X path implements the default HPTR lane-mode traffic-light encoder")
        self.hidden_dim, self.tl_state_dim, self.temp_window_size = hidden_dim, tl_state_dim, temp_window_size
        self.tl_lane_detach_mp_feature = tl_lane_detach_mp_feature
        self.register_buffer("hist_ohe", torch.eye(temp_window_size))
        self.temp_encoder = PolylineEncoder(hidden_dim=hidden_dim, tf_cfg=tf_cfg, **temp_encoder)
        self.n_tgt_knn_tl2tl = int(n_tgt_knn * k_tgt_knn_tl2tl)
        self.n_tgt_knn_tl2mp = int(n_tgt_knn * k_tgt_knn_tl2mp)
        self.dist_limit = dist_limit * k_dist_limit
        self.pose_rpe = pose_rpe
        self.tf_tl2tlmp = TransformerBlockRPE(n_layer=n_layer_tf, mode="dec_cross_attn", d_rpe=pose_rpe.out_dim, **tf_cfg)
        self.input_encoder = InputEncoder(hidden_dim=hidden_dim, attr_dim=tl_state_dim + temp_window_size, pe_dim=hidden_dim,
                                          **input_encoder)

    def pre_compute(self, tl_valid: Tensor, tl_attr: Tensor, tl_pose: Tensor, mp_token_invalid: Tensor,
                    mp_token_feature: Tensor, mp_token_pose: Tensor, mp_batch_div: int = 1, **kwargs) -> Dict[str, Tensor]:
        """tl_valid [n,L] bool, tl_attr [n,L] int64 lane index, tl_pose [n,L,3]; map tokens [n/div, M, ...]."""
        n, L = tl_valid.shape
        M = mp_token_pose.shape[1]
        dev = tl_pose.device
        rp = self.pose_rpe
        tl_inv = (~tl_valid).to(torch.uint8).contiguous()
        mp_inv = mp_token_invalid.to(torch.uint8).contiguous()
        mp_feat = mp_token_feature.reshape(-1, self.hidden_dim)
        bsel = (torch.arange(n, device=dev) // mp_batch_div).unsqueeze(1)
        t = {"tl_token_valid": tl_valid, "tl_token_invalid": ~tl_valid, "tl_token_invalid_u8": tl_inv,
             "tl_token_pose": tl_pose.float().contiguous(),
             "tl_token_attr": mp_token_feature[bsel, tl_attr].contiguous()}
        pose = t["tl_token_pose"]
        mp_pose = mp_token_pose.float().contiguous()
        # Static per scene: the 12-byte relative poses; the attention kernels rebuild the 128-d embedding from them as they do for the
        # agents' pairs. (Rounds 2-3 materialised the embeddings for few rollouts - 512 B per pair per layer and, in the one-launch
        # decoder layer, the run-time-mode instantiation with its spilled registers: configs[1] 470 -> 494 k with the poses instead.
        # TBX_TL_MAT_ROWS > n * L restores it.)
        mat = n * L < int(os.environ.get("TBX_TL_MAT_ROWS", 0))
        i_tt, m_tt, r_tt, e_tt = hip.knn_embed(pose, tl_inv, pose, tl_inv, self.n_tgt_knn_tl2tl, self.dist_limit, rp.pe_xy.freqs,
                                               rp.pe_yaw.freqs, rp.out_dim, want_rel_pose=not mat, want_emb=mat)
        i_tm, m_tm, r_tm, e_tm = hip.knn_embed(pose, tl_inv, mp_pose, mp_inv, self.n_tgt_knn_tl2mp, self.dist_limit, rp.pe_xy.freqs,
                                               rp.pe_yaw.freqs, rp.out_dim, tgt_batch_div=mp_batch_div, want_rel_pose=not mat,
                                               want_emb=mat)
        t.update(knn_idx_tl2tl=i_tt, knn_invalid_tl2tl=m_tt, rpe_tl2tl=e_tt, rel_tl2tl=r_tt, knn_idx_tl2mp=i_tm,
                 knn_invalid_tl2mp=m_tm, rpe_tl2mp=e_tm, rel_tl2mp=r_tm, mp_batch_div=mp_batch_div, n_mp=M, mp_feat_flat=mp_feat)
        return t

    def _kv_mp(self, t: Dict[str, Tensor], refresh: bool = False) -> Tensor:
        """K/V tables of the map tokens for this encoder's tl2mp layers (static per scene, cached in the token dict; refresh:
        recomputed into the cached tensor after the token features were overwritten in place)."""
        cache = t.setdefault("_kv_mp", {})
        key = (id(self), kv_dtype())  # (a table per element type: fp32 and bf16-table engines may share one token dict)
        if key not in cache or refresh:
            cache[key] = kv_tables(t["mp_feat_flat"].contiguous(), [(l.norm_tgt, l.attn) for l in self.tf_tl2tlmp.layers],
                                   out=cache.get(key))
        return cache[key]

    def _window_tile_images(self):
        """(input MLP images, PointNet images) for tbx_window_tile in "add" mode, or None where the module is not of the default shape
        (input encoder "add": d_in <= 32 -> 128 -> 128 -> 128 without layernorm; three 128 -> 64 PointNet layers)."""
        ie, te = self.input_encoder, self.temp_encoder
        lins = ie.mlp.linear_layers()
        if not (ie.mode == "add" and len(lins) == 3 and ie.mlp.output_dim == 128 and ie.mlp.input_dim <= 32 and self.hidden_dim == 128
                and all(ln is None for _, ln, _ in lins) and [a for _, _, a in lins] == [True, True, False] and len(te.mlp_layers) == 3):
            return None
        pn = []
        for mlp in te.mlp_layers:
            ll = mlp.linear_layers()
            if len(ll) != 1 or ll[0][1] is not None or tuple(ll[0][0].weight.shape) != (64, 128):
                return None
            pn.append(hip.packed_weight(ll[0][0].weight, ll[0][0].bias, mfma32=True))
        ins = [hip.packed_weight(hip.padded_weight(lins[0][0].weight, 32), lins[0][0].bias, mfma32=True)]
        ins += [hip.packed_weight(l.weight, l.bias, mfma32=True) for l, _, _ in lins[1:]]
        return ins, pn

    def prep_buffers(self, n: int, L: int, dev):
        """(attr [n*L*W, 16 | 32] f32, row_invalid [n*L*W] u8): what tbx_tl_prep writes for `encode`."""
        rows = n * L * self.temp_window_size
        ld_attr = 16 if 5 + self.temp_window_size <= 16 else 32
        return torch.empty(rows, ld_attr, dtype=torch.float32, device=dev), torch.empty(rows, dtype=torch.uint8, device=dev)

    def encode(self, hist_tl: Tensor, t: Dict[str, Tensor], tail: Optional[Callable[[Chain], None]] = None, prepared=None,
               tail_mf=None) -> Tensor:
        """hist_tl [n,L,W] u8 state masks (0xFF = step not yet seen), oldest first -> tl_token_feature [n*L, d].
        prepared = the prep_buffers() already filled for this window (by the launch that updated the lights, hip.sim_step(tl_prep=))."""
        n, L, W = hist_tl.shape
        assert W == self.temp_window_size
        dev, d = hist_tl.device, self.hidden_dim
        rows = n * L * W
        if prepared is not None:
            attr, row_inv = prepared
        else:
            attr, row_inv = self.prep_buffers(n, L, dev)
            hip.tl_prep(hist_tl, t["tl_token_invalid_u8"], attr, row_inv)
        x = torch.empty(n * L, d, dtype=torch.float32, device=dev)
        fp = first_proj_buffers(n * L, dev, hip.group_tile_rows(W, n * L))  # small launches: layer 0's projections in the windows' launch too
        wt_ = self._window_tile_images() if (fp is None and W <= 16 and (tile_rows_ok(n * L) or tile_small_ok())) else None
        if wt_ is not None and front_ok(n * L):  # ... and the block's first projection of the pooled rows in the same launch (tbx_front)
            l0 = self.tf_tl2tlmp.layers[0]
            dec = self.tf_tl2tlmp.mode == "dec_cross_attn"
            fp = front_proj_buffers(n * L, dev)
            hip.front(window=dict(attr=attr, pe=t["tl_token_attr"].reshape(n * L, d), row_invalid=row_inv, in_images=wt_[0], pn_images=wt_[1],
                                  window=W, out=x, add_mode=True),
                      proj=tile_proj_part(l0.norm_src if dec else l0.norm1, l0.attn_src if dec else l0.attn, fp["qkv"], True, fp["kv16"]))
        elif wt_ is not None:  # the whole temporal PointNet as one tbx_window_tile launch ("add" mode: + the light's lane feature)
            hip.window_tile(attr, t["tl_token_attr"].reshape(n * L, d), row_inv, wt_[0], wt_[1], W, x, add_mode=True)
        else:
            ch = Chain(hip.group_tile_rows(W, n * L), d + 4 if fp is None else FIRST_PROJ_LDW)
            cur = self.input_encoder.emit(ch, attr, t["tl_token_attr"].reshape(n * L, d), pe_row_div=W)
            kept = emit_pointnet(ch, self.temp_encoder, row_inv, x, x_buf=cur, keep=fp is not None)
            if fp is not None:
                emit_first_proj(ch, self.tf_tl2tlmp, fp, kept)
            ch.run(rows, group_rows=W)
        kv = self._kv_mp(t)
        M, div = t["n_mp"], t["mp_batch_div"]
        knn = SelfKnn(t["knn_idx_tl2tl"], t["knn_invalid_tl2tl"], t["rpe_tl2tl"], rel=t["rel_tl2tl"])
        run_block(self.tf_tl2tlmp, x, t["tl_token_invalid_u8"], n, L, knn,
                  cross=lambda l: [Seg(kv, l * 2 * D, l * 2 * D + D, M, t["knn_idx_tl2mp"], t["knn_invalid_tl2mp"],
                                       t["rpe_tl2mp"], div, rel=t["rel_tl2mp"])], tail=tail, pose_rpe=self.pose_rpe, first_proj=fp,
                  tail_mf=tail_mf)  # (tail_mf: the same tail as tbx_tl_tail_t fields, for the last layer's one-launch form)
        return x

    @staticmethod
    def states_to_hist(tl_state: Tensor, window: int) -> Tensor:
        """[n,L,n_step,5] one-hot bool -> [n,L,window] u8 bit masks, left-padded with 0xFF (missing)."""
        n, L, n_step, S = tl_state.shape
        assert n_step <= window
        bits = (tl_state.to(torch.int32) << torch.arange(S, device=tl_state.device, dtype=torch.int32)).sum(-1).to(torch.uint8)
        pad = torch.full((n, L, window - n_step), 0xFF, dtype=torch.uint8, device=tl_state.device)
        return torch.cat([pad, bits], 2).contiguous()

    def forward(self, tl_state: Tensor, called_by_latent_encoder: bool = False, **tl_tokens) -> Tensor:
        """Reference signature: tl_state [n,L,n_step,5] + the pre_compute dict -> [n,L,d]."""
        n, L = tl_state.shape[:2]
        return self.encode(self.states_to_hist(tl_state, self.temp_window_size), tl_tokens).view(n, L, self.hidden_dim)


class TrafficLightStatePredictor(nn.Module):
    def __init__(self, hidden_dim: int, tl_state_dim: int, n_layer: int, rnn_dropout_p: float, temp_window_size: int,
                 detach_tl_feature: bool) -> None:
        super().__init__()
        if temp_window_size <= 0:
            raise NotImplementedError("RNN variant is not on the default hot path")
        self.temp_window_size, self.detach_tl_feature = temp_window_size, detach_tl_feature
        self.hidden_dim, self.tl_state_dim = hidden_dim, tl_state_dim
        self.mlp = MLP([hidden_dim] * n_layer + [tl_state_dim], end_layer_activation=False)

    def init(self) -> None:
        pass

    def emit(self, ch: Chain, tl_invalid_u8: Tensor, out_logits: Tensor):
        """x in BUF1[:, 0:d] -> clamp(MLP(x) masked, -3, 3) stored to out_logits [rows, 5]."""
        d = self.hidden_dim
        lins = [t[0] for t in self.mlp.linear_layers()]
        assert len(lins) == 3
        ch.linear(BUF1, 0, BUF0, 0, lins[0].weight, lins[0].bias, relu=True)
        ch.linear(BUF0, 0, BUF0, d, lins[1].weight, lins[1].bias, relu=True)
        ch.linear(BUF0, d, BUF0, 2 * d, lins[2].weight, lins[2].bias)
        ch.rowmask(BUF0, 2 * d, self.tl_state_dim, mask=tl_invalid_u8)
        ch.clamp(BUF0, 2 * d, self.tl_state_dim, -3.0, 3.0)
        ch.store(BUF0, 2 * d, self.tl_state_dim, out_logits)

    def forward(self, tl_token_feature: Tensor, tl_token_invalid: Tensor) -> Tensor:
        n, L, d = tl_token_feature.shape
        x = tl_token_feature.reshape(n * L, d).contiguous().float()
        out = torch.empty(n * L, self.tl_state_dim, dtype=torch.float32, device=x.device)
        ch = Chain(16, 3 * d + 4)
        ch.load(x, BUF1, 0, n=d)
        self.emit(ch, tl_token_invalid.reshape(-1).to(torch.uint8).contiguous(), out)
        ch.run(n * L)
        return out.view(n, L, self.tl_state_dim)
