"""`MapEncoder` (models/map_encoder.py:14-113): polyline PointNet + 8 layers of KNARPE self-attention over map
tokens. Runs once per scene; every stage is a HIP kernel (tbx_map_prep, tbx_rowchain, tbx_knn_embed, tbx_knarpe_attn)."""
from typing import Dict

import torch
from torch import Tensor, nn

from .. import hip
from ..engine import SelfKnn, emit_pointnet, run_block
from ..hip import Chain
from ..utils.pose_emb import PoseEmb
from .modules.input_encoder import InputEncoder
from .modules.polyline_encoder import PolylineEncoder
from .modules.transformer_rpe import TransformerBlockRPE


class MapEncoder(nn.Module):
    def __init__(self, hidden_dim: int, attr_dim: int, pairwise_relative: bool, pose_emb, n_mp_pl_node: int, input_encoder,
                 pl_encoder, pose_rpe: nn.Module, tf_cfg, n_layer_tf: int, n_tgt_knn: int, dist_limit: float) -> None:
        super().__init__()
        if not pairwise_relative or input_encoder["mode"] != "cat":
            raise NotImplementedError("the MI355X path implements the pairwise-relative HPTR map encoder")
        self.pairwise_relative, self.pose_rpe = pairwise_relative, pose_rpe
        self.n_tgt_knn, self.dist_limit, self.hidden_dim, self.n_node = n_tgt_knn, dist_limit, hidden_dim, n_mp_pl_node
        self.register_buffer("pl_node_ohe", torch.eye(n_mp_pl_node)[None, None, :, :])
        self.pose_emb = PoseEmb(pe_dim=hidden_dim // 2, **pose_emb)
        self.input_encoder = InputEncoder(hidden_dim=hidden_dim, attr_dim=attr_dim + n_mp_pl_node,
                                          pe_dim=self.pose_emb.out_dim, **input_encoder)
        self.pl_encoder = PolylineEncoder(hidden_dim=hidden_dim, tf_cfg=tf_cfg, **pl_encoder)
        self.tf_mp2mp = TransformerBlockRPE(n_layer=n_layer_tf, mode="enc_self_attn", d_rpe=self.pose_rpe.out_dim, **tf_cfg)

    def forward(self, mp_valid: Tensor, mp_attr: Tensor, mp_pose: Tensor, mp_type: Tensor) -> Dict[str, Tensor]:
        """mp_valid [n,M,N] bool, mp_attr [n,M,11] float, mp_pose [n,M,N,3], mp_type [n,M,11] one-hot bool."""
        n, M, N = mp_valid.shape
        dev, d = mp_pose.device, self.hidden_dim
        rows = n * M * N
        attr = torch.empty(rows, 32, dtype=torch.float32, device=dev)
        pe = torch.empty(rows, 8, dtype=torch.float32, device=dev)
        row_inv = torch.empty(rows, dtype=torch.uint8, device=dev)
        tok_pose = torch.empty(n, M, 3, dtype=torch.float32, device=dev)
        tok_inv = torch.empty(n, M, dtype=torch.uint8, device=dev)
        hip.map_prep(mp_valid.to(torch.uint8).contiguous(), mp_attr.float().contiguous(), mp_pose.float().contiguous(), attr, pe,
                     row_inv, tok_pose, tok_inv)
        feat = torch.empty(n * M, d, dtype=torch.float32, device=dev)
        ch = Chain(16 if N <= 16 else 32, d + 4)
        cur = self.input_encoder.emit(ch, attr, pe)
        emit_pointnet(ch, self.pl_encoder, row_inv, feat, x_buf=cur)
        ch.run(rows, group_rows=N)
        idx, inv, rel, _ = hip.knn_embed(tok_pose, tok_inv, tok_pose, tok_inv, self.n_tgt_knn, self.dist_limit,
                                         want_rel_pose=True, want_emb=False)
        run_block(self.tf_mp2mp, feat, tok_inv, n, M, SelfKnn(idx, inv, rel=rel), pose_rpe=self.pose_rpe)
        return {"mp_token_invalid": tok_inv.bool(), "mp_token_feature": feat.view(n, M, d), "mp_token_pose": tok_pose,
                "mp_token_type": mp_type, "knn_idx_mp2mp": idx, "knn_invalid_mp2mp": inv}
