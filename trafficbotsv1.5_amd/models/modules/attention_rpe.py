"""`AttentionRPE` = KNARPE (modules/attention_rpe.py:10-198), rpe branch, on the fused HIP kernel."""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from ... import hip
from ...engine import D, O_LD, Q_LD, emit_attn_out, emit_qkv
from ...hip import AUX, BUF0, BUF1, Chain, Seg


class AttentionRPE(nn.Module):
    def __init__(self, d_model: int, n_head: int, dropout_p: float = 0.1, bias: bool = True, d_rpe: int = -1,
                 apply_q_rpe: bool = False) -> None:
        super().__init__()
        if apply_q_rpe or not bias:
            raise NotImplementedError("apply_q_rpe / bias=False are not on the default hot path")
        self.d_model, self.n_head, self.d_head, self.d_rpe, self.apply_q_rpe = d_model, n_head, d_model // n_head, d_rpe, apply_q_rpe
        assert self.d_head * n_head == d_model
        if d_rpe > 0:
            self.linear_rpe = nn.Linear(d_rpe, 2 * d_model, bias=bias)
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.out_proj_weight = nn.Parameter(torch.empty(d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj_bias = nn.Parameter(torch.zeros(d_model))
        self.dropout_p = dropout_p
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj_weight)

    def forward(self, src: Tensor, tgt: Optional[Tensor] = None, tgt_padding_mask: Optional[Tensor] = None,
                attn_mask: Optional[Tensor] = None, rpe: Optional[Tensor] = None, need_weights=False
                ) -> Tuple[Tensor, Optional[Tensor]]:
        """src [n,S,d]; tgt [n,S,K,d] (gathered KNN targets); tgt_padding_mask [n,S,K]; rpe [n,S,K,d_rpe]."""
        if tgt is None or tgt.dim() != 4 or rpe is None or attn_mask is not None or need_weights:
            raise NotImplementedError("only the KNN + rpe branch of AttentionRPE is on the hot path")
        if (self.d_model, self.n_head, self.d_rpe) != (128, 4, 128):
            raise NotImplementedError("the gfx950 kernel is built for d_model=128, n_head=4, d_rpe=128")
        n, S, K, d = tgt.shape
        dev = src.device
        x = src.reshape(n * S, d).contiguous().float()
        if self.training:  # attention_rpe.py:83-198 in train mode: autograd + keyed probability dropout (train_graph.attention)
            from ... import train_graph as TG

            idx = torch.arange(S * K, dtype=torch.int32, device=dev).view(1, S, K).expand(n, -1, -1).contiguous()
            t = TG.Targets(tgt.reshape(n * S * K, d).contiguous().float(), idx, tgt_padding_mask.to(torch.uint8).contiguous(),
                           rpe.contiguous().float(), S * K)
            with TG.module_scope(n, dev):
                y = TG.attention(self, x, [t], [TG.kv_table(self, None, t)], n, S)
            return y.view(n, S, d), None
        # per-pair K/V projection (what the reference does): a table with one row per (src, tgt) pair
        t2 = tgt.reshape(n * S * K, d).contiguous().float()
        kv = torch.empty(n * S * K, 2 * d, dtype=torch.float32, device=dev)
        ch = Chain(16, 388)
        ch.load(t2, BUF0, 0, n=d)
        ch.linear(BUF0, 0, BUF1, 0, self.in_proj_weight[d:], self.in_proj_bias[d:])
        ch.store(BUF1, 0, 2 * d, kv)
        ch.run(t2.shape[0])
        q = torch.empty(n * S, Q_LD, dtype=torch.float32, device=dev)
        ch = Chain(16, 772)
        ch.load(x, BUF0, 0, n=d)
        w = emit_qkv(ch, self, BUF0, 0, BUF0, D, with_kv=False)
        ch.store(BUF0, D, w, q)
        ch.run(n * S)
        idx = torch.arange(S * K, dtype=torch.int32, device=dev).view(1, S, K).expand(n, -1, -1).contiguous()
        inv = tgt_padding_mask.to(torch.uint8).contiguous()
        obuf = torch.empty(n * S, O_LD, dtype=torch.float32, device=dev)
        flag = torch.empty(n * S, dtype=torch.uint8, device=dev)
        hip.knarpe_attn(q, 0, D, self.linear_rpe.bias, n, S, [Seg(kv, 0, d, S * K, idx, inv, rpe.contiguous().float())], obuf, flag)
        out = torch.empty(n * S, d, dtype=torch.float32, device=dev)
        ch = Chain(16, 644)
        ch.zero(BUF1, 0, d)
        emit_attn_out(ch, self, obuf, flag)
        ch.store(BUF1, 0, d, out)
        ch.run(n * S)
        return out.view(n, S, d), None
