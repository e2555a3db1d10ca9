"""`MLP` with the reference's constructor and state-dict layout (modules/mlp.py:20-72); forward runs as one
tbx_rowchain program (Linear [+LayerNorm] + ReLU stages on exact-fp32 MFMA)."""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from ... import hip
from ...engine import emit_mlp
from ...hip import BUF0, BUF1, Chain


class MLP(nn.Module):
    def __init__(self, fc_dims: Sequence[int], dropout_p: float = -1.0, activation: str = "relu",
                 end_layer_activation: bool = True, init_weight_norm: bool = False, init_bias: Optional[float] = None,
                 use_layernorm: bool = False, use_batchnorm: bool = False) -> None:
        super().__init__()
        if activation != "relu" or use_batchnorm:
            raise NotImplementedError("the MI355X path implements the default relu / no-batchnorm MLP")
        assert len(fc_dims) >= 2
        seq: List[nn.Module] = []
        n_lin = len(fc_dims) - 1
        for i in range(n_lin):
            fc = nn.Linear(fc_dims[i], fc_dims[i + 1])
            if init_weight_norm:
                fc.weight.data *= 1.0 / fc.weight.norm(dim=1, p=2, keepdim=True)
            if init_bias is not None and i == n_lin - 1:
                fc.bias.data.fill_(init_bias)
            seq.append(fc)
            if i < n_lin - 1 or end_layer_activation:
                if use_layernorm:
                    seq.append(nn.LayerNorm(fc_dims[i + 1]))
                seq.append(nn.ReLU(inplace=True))
            if dropout_p > 0:
                seq.append(nn.Dropout(p=dropout_p))
        self.input_dim, self.output_dim = fc_dims[0], fc_dims[-1]
        self.dropout_p = dropout_p
        self.fc_layers = nn.Sequential(*seq)

    def linear_layers(self) -> List[Tuple[nn.Linear, Optional[nn.LayerNorm], bool]]:
        """[(Linear, LayerNorm | None, followed_by_relu)] in execution order."""
        mods = list(self.fc_layers)
        out = []
        for i, m in enumerate(mods):
            if isinstance(m, nn.Linear):
                ln, act, j = None, False, i + 1
                while j < len(mods) and not isinstance(mods[j], nn.Linear):
                    if isinstance(mods[j], nn.LayerNorm):
                        ln = mods[j]
                    elif isinstance(mods[j], nn.ReLU):
                        act = True
                    j += 1
                out.append((m, ln, act))
        return out

    def forward(self, x: Tensor, mask_invalid: Optional[Tensor] = None, fill_invalid: float = 0.0) -> Tensor:
        if self.training:  # mlp.py:58-72 in train mode: library GEMMs + autograd, keyed dropout (train_graph.mlp)
            from ... import train_graph as TG

            lead0 = x.shape[:-1]
            x0 = x.reshape(-1, self.input_dim).contiguous().float()
            with TG.module_scope(1, x.device):
                y = TG.mlp(self, x0, True)
            if mask_invalid is not None:
                y = y.masked_fill(mask_invalid.reshape(-1, 1).bool(), fill_invalid)
            return y.view(*lead0, self.output_dim)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.input_dim).contiguous().float()
        rows = x2.shape[0]
        width = max([self.input_dim] + [l.weight.shape[0] for l, _, _ in self.linear_layers()])
        ldw = ((width + 15) // 16) * 16 + 4
        out = torch.empty(rows, self.output_dim, dtype=torch.float32, device=x.device)
        ch = Chain(16, ldw)
        ch.load(x2, BUF0, 0, n=self.input_dim, pad_to=((self.input_dim + 15) // 16) * 16)
        cur = emit_mlp(ch, self, BUF0, 0)
        if mask_invalid is not None:
            ch.rowmask(cur, 0, self.output_dim, mask=mask_invalid.reshape(-1).to(torch.uint8).contiguous(), fill=fill_invalid)
        ch.store(cur, 0, self.output_dim, out)
        ch.run(rows)
        return out.view(*lead, self.output_dim)
