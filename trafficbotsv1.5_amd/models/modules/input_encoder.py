"""`InputEncoder` (modules/input_encoder.py:8-61): MLP(attr) combined with a positional feature by cat / add."""
from typing import Optional

import torch
from torch import Tensor, nn

from ...engine import emit_mlp
from ...hip import BUF0, BUF1, Chain
from .mlp import MLP


class InputEncoder(nn.Module):
    def __init__(self, hidden_dim: int, attr_dim: int, pe_dim: int, n_layer: int, mlp_dropout_p: float,
                 mlp_use_layernorm: bool, mode: str) -> None:
        super().__init__()
        self.mode, self.hidden_dim, self.pe_dim = mode, hidden_dim, pe_dim
        if mode == "input":
            d_in, d_out = attr_dim + pe_dim, hidden_dim
        elif mode == "cat":
            d_in, d_out = attr_dim, hidden_dim - pe_dim
            assert d_out >= 32, f"Make sure pe_dim is smaller than {hidden_dim - 32}!"
        elif mode == "add":
            d_in, d_out = attr_dim, hidden_dim
            assert pe_dim in (0, hidden_dim)
        else:
            raise NotImplementedError(mode)
        self.mlp = MLP([d_in] + [d_out] * n_layer, dropout_p=mlp_dropout_p, use_layernorm=mlp_use_layernorm,
                       end_layer_activation=False)

    def emit(self, ch: Chain, attr: Tensor, pe: Optional[Tensor], pe_row_div: int = 0, attr_ld_pad: Optional[int] = None) -> int:
        """Stages that leave the [.., hidden_dim] feature in the returned buffer at column 0.
        attr: [rows, >= attr_dim] (zero padded to a multiple of 16 columns if wider), pe: [rows(/div), pe_dim]."""
        d_in = self.mlp.input_dim
        pad = ((d_in + 15) // 16) * 16
        ch.load(attr, BUF0, 0, n=d_in, pad_to=pad)
        cur = emit_mlp(ch, self.mlp, BUF0, 0)
        if pe is not None:
            if self.mode == "cat":
                ch.load(pe, cur, self.mlp.output_dim, n=self.pe_dim, row_div=pe_row_div)
            elif self.mode == "add":
                ch.load(pe, cur, 0, n=self.hidden_dim, accum=True, row_div=pe_row_div)
            else:
                raise NotImplementedError("mode=input is not on the default hot path")
        return cur

    def forward(self, attr: Tensor, pe: Optional[Tensor]) -> Tensor:
        lead = attr.shape[:-1]
        a2 = attr.reshape(-1, attr.shape[-1]).contiguous().float()
        p2 = None if pe is None else pe.reshape(-1, pe.shape[-1]).contiguous().float()
        out = torch.empty(a2.shape[0], self.hidden_dim, dtype=torch.float32, device=attr.device)
        ch = Chain(16, 132 + 16)
        cur = self.emit(ch, a2, p2)
        ch.store(cur, 0, self.hidden_dim, out)
        ch.run(a2.shape[0])
        return out.view(*lead, self.hidden_dim)
