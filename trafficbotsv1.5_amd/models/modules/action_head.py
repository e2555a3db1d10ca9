"""`ActionHead` (modules/action_head.py:9-100), branch_type=True with per-type log_std parameters."""
from typing import Optional

import torch
from torch import Tensor, nn
from torch.distributions import Independent, Normal

from ... import hip
from ...hip import AUX, BUF0, BUF1, Chain
from .mlp import MLP


class ActionHead(nn.Module):
    def __init__(self, hidden_dim: int, action_dim: int, n_layer: int, mlp_use_layernorm: bool,
                 log_std: Optional[float] = None, branch_type: bool = False, n_ag_type: int = 3) -> None:
        super().__init__()
        if not branch_type or log_std is None or mlp_use_layernorm:
            raise NotImplementedError("the MI355X path implements the default branch_type head with fixed log_std")
        self.branch_type, self.out_dim, self.hidden_dim = branch_type, action_dim, hidden_dim
        self.mlp_mean = nn.ModuleList([MLP([hidden_dim] * n_layer + [action_dim], end_layer_activation=False)
                                       for _ in range(n_ag_type)])
        self.log_std = nn.ParameterList([nn.Parameter(log_std * torch.ones(action_dim)) for _ in range(n_ag_type)])
        self.fused_branches = True  # emit(): the branches as 3 stacked / block-diagonal stages (False: 9 per-branch stages)
        self.masked_sum_store = True  # ... and their masked sum in the storing stage (False: ROWMASK + COPY / ADD per branch)

    def emit(self, ch: Chain, type_mask: Tensor, out_mean: Tensor):
        """x in BUF1[:, 0:d]; type_mask u8 [3, rows] = ~(type_i & valid); writes the masked-sum mean to out_mean."""
        d = self.hidden_dim
        G = len(self.mlp_mean)
        lins = [[t[0] for t in mlp.linear_layers()] for mlp in self.mlp_mean]
        assert all(len(l) == 3 for l in lins)
        if self.fused_branches and ch.pack_weights and G > 1 and min(ch.ldw, ch.ldw1) >= G * d and self.out_dim <= 16 and d % 16 == 0:
            # the per-type branches as THREE stages instead of nine: layer 1 of every branch reads the same x (one d -> G*d
            # stage), layers 2 and 3 are block-diagonal stages (groups = G; the 2-wide outputs zero-padded to 16 rows per
            # branch). Column for column the same sums in the same order as the per-branch stages: bit-identical. x (BUF1) is
            # dead after layer 1, so layer 2 lands there.
            w1, b1 = hip.stacked_linear([l[0] for l in lins])
            w2, b2 = hip.stacked_linear([l[1] for l in lins])
            w3, b3 = hip.stacked_linear([l[2] for l in lins], pad_out_to=16)
            ch.linear(BUF1, 0, BUF0, 0, w1, b1, relu=True)
            ch.linear(BUF0, 0, BUF1, 0, w2, b2, relu=True, groups=G, src_stride=d, dst_stride=d)
            ch.linear(BUF1, 0, BUF0, 0, w3, b3, groups=G, src_stride=d, dst_stride=16)
            if self.masked_sum_store and type_mask.is_contiguous():
                # the masked sum over the branches in the storing stage (TBX_F_MASKED_SUM): 1 stage instead of 2 G + 1, same sums
                ch.store_masked_sum(BUF0, 0, self.out_dim, 16, type_mask, out_mean)
                return
            for i in range(G):
                ch.rowmask(BUF0, 16 * i, self.out_dim, mask=type_mask[i])
                (ch.copy if i == 0 else ch.add)(BUF0, 16 * i, AUX, 0, self.out_dim)
            ch.store(AUX, 0, self.out_dim, out_mean)
            return
        for i in range(G):
            l = lins[i]
            ch.linear(BUF1, 0, BUF0, 0, l[0].weight, l[0].bias, relu=True)
            ch.linear(BUF0, 0, BUF0, d, l[1].weight, l[1].bias, relu=True)
            ch.linear(BUF0, d, BUF0, 2 * d, l[2].weight, l[2].bias)
            ch.rowmask(BUF0, 2 * d, self.out_dim, mask=type_mask[i])
            (ch.copy if i == 0 else ch.add)(BUF0, 2 * d, AUX, 0, self.out_dim)
        ch.store(AUX, 0, self.out_dim, out_mean)

    def masked_log_std(self, valid: Tensor, ag_type: Tensor) -> Tensor:
        m = (ag_type & valid.unsqueeze(-1)).unsqueeze(-1).to(self.log_std[0].dtype)  # [n,A,3,1]
        return (m * torch.stack(list(self.log_std), 0)).sum(2)

    def forward(self, x: Tensor, valid: Tensor, ag_type: Tensor) -> Independent:
        n_sc, n_ag, _ = ag_type.shape
        x2 = x.reshape(-1, self.hidden_dim).contiguous().float()
        tm = (~(ag_type & valid.unsqueeze(-1))).permute(2, 0, 1).reshape(3, -1).to(torch.uint8).contiguous()
        mean = torch.empty(x2.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        ch = Chain(16, 3 * self.hidden_dim + 4)
        ch.load(x2, BUF1, 0, n=self.hidden_dim)
        self.emit(ch, tm, mean)
        ch.run(x2.shape[0])
        return Independent(Normal(mean.view(n_sc, n_ag, self.out_dim), self.masked_log_std(valid, ag_type).exp()), 1)
