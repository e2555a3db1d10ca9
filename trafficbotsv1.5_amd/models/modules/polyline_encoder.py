"""`PolylineEncoder` (modules/polyline_encoder.py:11-63), PointNet variant: one workgroup per polyline / track."""
import torch
from torch import Tensor, nn

from ...engine import emit_pointnet
from ...hip import BUF1, Chain
from .mlp import MLP


class PolylineEncoder(nn.Module):
    def __init__(self, hidden_dim: int, tf_cfg, n_layer: int, mlp_use_layernorm: bool, mlp_dropout_p: float,
                 use_pointnet: bool, pooling_mode: str) -> None:
        super().__init__()
        if not use_pointnet or pooling_mode != "max_valid" or mlp_use_layernorm:
            raise NotImplementedError("the MI355X path implements the default PointNet / max_valid polyline encoder")
        self.use_pointnet, self.pooling_mode, self.hidden_dim = use_pointnet, pooling_mode, hidden_dim
        self.mlp_dropout_p = mlp_dropout_p
        self.tile_rows = None  # None: one group per tile (16 / 32 rows by n_node); 32: floor(32 / n_node) groups per tile
        self.mlp_layers = nn.ModuleList([MLP([hidden_dim, hidden_dim // 2], dropout_p=mlp_dropout_p) for _ in range(n_layer)])

    def forward(self, x: Tensor, invalid: Tensor) -> Tensor:
        """x [n_sc, n_mp, n_node, d], invalid [n_sc, n_mp, n_node] -> [n_sc, n_mp, d]."""
        n_sc, n_mp, n_node, d = x.shape
        if self.training:  # polyline_encoder.py:49-61 in train mode (train_graph.pointnet: fused relu / dropout / masked max + autograd)
            from ... import train_graph as TG

            with TG.module_scope(n_sc, x.device):
                y = TG.pointnet(self, x.reshape(n_sc * n_mp, n_node, d).contiguous().float(), invalid.reshape(n_sc * n_mp, n_node).bool(), True)
            return y.view(n_sc, n_mp, d)
        x2 = x.reshape(-1, d).contiguous().float()
        inv = invalid.reshape(-1).to(torch.uint8).contiguous()
        out = torch.empty(n_sc * n_mp, d, dtype=torch.float32, device=x.device)
        ch = Chain(self.tile_rows or (16 if n_node <= 16 else 32), d + 4)
        ch.load(x2, BUF1, 0, n=d)
        emit_pointnet(ch, self, inv, out)
        ch.run(x2.shape[0], group_rows=n_node)
        return out.view(n_sc, n_mp, d)
