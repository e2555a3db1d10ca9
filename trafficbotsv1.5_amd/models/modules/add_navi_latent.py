"""`AddNaviLatent` (modules/add_navi_latent.py:8-65), mode=cat, res_add=True: x + MLP([x, MLP_in(z)])."""
from typing import Optional

import torch
from torch import Tensor, nn

from ...hip import BUF0, BUF1, Chain
from .mlp import MLP


class AddNaviLatent(nn.Module):
    def __init__(self, hidden_dim: int, in_dim: int, dummy: bool, mode: str, n_layer: int, mlp_use_layernorm: bool,
                 mlp_dropout_p: float, res_add: bool = False) -> None:
        super().__init__()
        self.dummy, self.hidden_dim, self.in_dim = dummy, hidden_dim, in_dim
        if not dummy:
            if mode != "cat" or not res_add or mlp_use_layernorm:
                raise NotImplementedError("the MI355X path implements the default cat / res_add AddNaviLatent")
            self.mode, self.res_add, self.mlp_dropout_p = mode, res_add, mlp_dropout_p
            self.mlp_in = MLP([in_dim] + [hidden_dim] * n_layer, dropout_p=mlp_dropout_p)
            self.mlp = MLP([2 * hidden_dim] + [hidden_dim] * n_layer, dropout_p=mlp_dropout_p)

    def emit_embed(self, ch: Chain, z: Tensor, out: Tensor, z_invalid: Optional[Tensor] = None) -> bool:
        """mlp_in(z) alone -> out [rows, d]: for inputs that stay the same over a whole rollout (the latent) the engine evaluates
        this once and hands the result to `emit(..., z_embedded=...)` every step. With z_invalid (fixed over the rollout too) the
        validity mask is applied here, by the last stage - returns True if it was (`emit(..., z_premasked=True)`)."""
        from ...engine import current as _sched
        d = self.hidden_dim
        l_in = [t[0] for t in self.mlp_in.linear_layers()]
        pad = ((self.in_dim + 15) // 16) * 16
        masked = z_invalid is not None and _sched().rowzero and ch.pack_weights
        ch.load(z, BUF0, 0, n=self.in_dim, pad_to=pad)
        ch.linear(BUF0, 0, BUF0, 2 * d, l_in[0].weight, l_in[0].bias, relu=True)
        ch.linear(BUF0, 2 * d, BUF0, d, l_in[1].weight, l_in[1].bias, relu=True)
        if masked:
            ch.linear(BUF0, d, BUF0, 2 * d, l_in[2].weight, l_in[2].bias, relu=True, skip_rows=z_invalid, zero_skipped=True)
        else:
            ch.linear(BUF0, d, BUF0, 2 * d, l_in[2].weight, l_in[2].bias, relu=True)
        ch.store(BUF0, 2 * d, d, out)
        return masked

    def emit_embed_buf(self, ch: Chain, out: Tensor, z_invalid: Optional[Tensor] = None, mask_is_valid: bool = False) -> bool:
        """mlp_in(z) for z already in BUF0[:, d:2d] -> out [rows, d]: the same three stages as in `emit(z=None)`, for callers that
        evaluate them ahead of the chain that owns x (inference: no dropout between them). With z_invalid the validity mask is
        applied by the last stage (TBX_F_ROWZERO) - returns True if it was (`emit(..., z_premasked=True)` then skips its ROWMASK)."""
        from ...engine import current as _sched
        d = self.hidden_dim
        l_in = [t[0] for t in self.mlp_in.linear_layers()]
        masked = z_invalid is not None and _sched().rowzero and ch.pack_weights
        ch.linear(BUF0, d, BUF0, 2 * d, l_in[0].weight, l_in[0].bias, relu=True)
        ch.linear(BUF0, 2 * d, BUF0, d, l_in[1].weight, l_in[1].bias, relu=True)
        if masked:
            ch.linear(BUF0, d, BUF0, 2 * d, l_in[2].weight, l_in[2].bias, relu=True, skip_rows=z_invalid, skip_is_valid=mask_is_valid,
                      zero_skipped=True)
        else:
            ch.linear(BUF0, d, BUF0, 2 * d, l_in[2].weight, l_in[2].bias, relu=True)
        ch.store(BUF0, 2 * d, d, out)
        return masked

    def emit(self, ch: Chain, z_invalid: Tensor, z: Optional[Tensor] = None, mask_is_valid: bool = False,
             z_embedded: Optional[Tensor] = None, z_premasked: bool = False):
        """x in BUF1[:, 0:d] (updated in place). z either already in BUF0[:, d:2d] (z=None) or loaded from `z`
        [rows, in_dim], or given as `z_embedded` = mlp_in(z) [rows, d] (emit_embed; same values, three stages fewer).
        Uses BUF0 columns [0, 4d)."""
        d = self.hidden_dim
        l_in, l_mlp = [t[0] for t in self.mlp_in.linear_layers()], [t[0] for t in self.mlp.linear_layers()]
        assert len(l_in) == 3 and len(l_mlp) == 3, "default n_layer = 3"
        if z_embedded is not None:
            from ...engine import current as _sched
            ch.load(z_embedded, BUF0, 2 * d, n=d)
            if not z_premasked:  # (the producer of z_embedded already zeroed the invalid rows)
                ch.rowmask(BUF0, 2 * d, d, mask=z_invalid, valid_mask=mask_is_valid)
            ch.copy(BUF1, 0, BUF0, d, d)  # [x | z] at BUF0[:, d:3d]
            ch.linear(BUF0, d, BUF0, 3 * d, l_mlp[0].weight, l_mlp[0].bias, relu=True)
            ch.linear(BUF0, 3 * d, BUF0, 0, l_mlp[1].weight, l_mlp[1].bias, relu=True)
            if _sched().rowzero and ch.pack_weights:  # the closing mask inside the last stage (relu, then 0 for the masked rows)
                ch.linear(BUF0, 0, BUF0, 3 * d, l_mlp[2].weight, l_mlp[2].bias, relu=True, skip_rows=z_invalid,
                          skip_is_valid=mask_is_valid, zero_skipped=True)
            else:
                ch.linear(BUF0, 0, BUF0, 3 * d, l_mlp[2].weight, l_mlp[2].bias, relu=True)
                ch.rowmask(BUF0, 3 * d, d, mask=z_invalid, valid_mask=mask_is_valid)
            ch.add(BUF0, 3 * d, BUF1, 0, d)
            return
        from ...engine import drop_site

        def drop(buf_col, mlp):  # the MLP's dropout after a layer (training's stepping pass; nothing in inference)
            dd = drop_site(mlp.dropout_p)
            if dd is not None:
                ch.dropout(BUF0, buf_col, d, *dd)

        if z is not None:
            pad = ((self.in_dim + 15) // 16) * 16
            ch.load(z, BUF0, 0, n=self.in_dim, pad_to=pad)
            ch.linear(BUF0, 0, BUF0, 2 * d, l_in[0].weight, l_in[0].bias, relu=True)
        else:
            ch.linear(BUF0, d, BUF0, 2 * d, l_in[0].weight, l_in[0].bias, relu=True)
        drop(2 * d, self.mlp_in)
        ch.linear(BUF0, 2 * d, BUF0, d, l_in[1].weight, l_in[1].bias, relu=True)
        drop(d, self.mlp_in)
        ch.linear(BUF0, d, BUF0, 2 * d, l_in[2].weight, l_in[2].bias, relu=True)
        drop(2 * d, self.mlp_in)
        ch.rowmask(BUF0, 2 * d, d, mask=z_invalid, valid_mask=mask_is_valid)
        ch.copy(BUF1, 0, BUF0, d, d)  # [x | z] at BUF0[:, d:3d]
        ch.linear(BUF0, d, BUF0, 3 * d, l_mlp[0].weight, l_mlp[0].bias, relu=True)
        drop(3 * d, self.mlp)
        ch.linear(BUF0, 3 * d, BUF0, 0, l_mlp[1].weight, l_mlp[1].bias, relu=True)
        drop(0, self.mlp)
        ch.linear(BUF0, 0, BUF0, 3 * d, l_mlp[2].weight, l_mlp[2].bias, relu=True)
        drop(3 * d, self.mlp)
        ch.rowmask(BUF0, 3 * d, d, mask=z_invalid, valid_mask=mask_is_valid)
        ch.add(BUF0, 3 * d, BUF1, 0, d)

    def forward(self, x: Tensor, z: Optional[Tensor], z_valid: Optional[Tensor] = None) -> Tensor:
        if self.dummy:
            return x
        if self.training:  # add_navi_latent.py:52-65 in train mode (train_graph.add_navi_latent)
            from ... import train_graph as TG

            lead0 = x.shape[:-1]
            x0 = x.reshape(-1, self.hidden_dim).contiguous().float()
            z0 = z.reshape(-1, self.in_dim).contiguous().float()
            zi0 = torch.zeros(x0.shape[0], dtype=torch.bool, device=x.device) if z_valid is None else (~z_valid).reshape(-1)
            with TG.module_scope(1, x.device):
                y = TG.add_navi_latent(self, x0, z0, zi0, True)
            return y.view(*lead0, self.hidden_dim)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.hidden_dim).contiguous().float()
        z2 = z.reshape(-1, self.in_dim).contiguous().float()
        zi = torch.zeros(x2.shape[0], dtype=torch.uint8, device=x.device) if z_valid is None else (~z_valid).reshape(-1).to(torch.uint8)
        out = torch.empty_like(x2)
        ch = Chain(16, 4 * self.hidden_dim + 4)
        ch.load(x2, BUF1, 0, n=self.hidden_dim)
        self.emit(ch, zi.contiguous(), z2)
        ch.store(BUF1, 0, self.hidden_dim, out)
        ch.run(x2.shape[0])
        return out.view(*lead, self.hidden_dim)
