"""`TransformerBlockRPE` / `TransformerRPE` (modules/transformer_rpe.py:20-245): pre-LN KNARPE transformer layers.
The layers are parameter containers; `TransformerBlockRPE.forward` schedules the whole stack through engine.run_block."""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from ...engine import D, SelfKnn, kv_tables, run_block
from ...hip import Seg
from .attention_rpe import AttentionRPE


class TransformerRPE(nn.Module):
    def __init__(self, d_model: int, n_head: int, k_feedforward: int, dropout_p: float, bias: bool, activation: str,
                 mode: str, d_rpe: int = -1, apply_q_rpe: bool = False) -> None:
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("relu is the default hot path activation")
        self.mode, self.dropout_p = mode, dropout_p
        self.norm1 = nn.LayerNorm(d_model)
        self.norm_tgt = nn.LayerNorm(d_model)
        if mode == "dec_cross_attn":
            self.attn_src = AttentionRPE(d_model, n_head, dropout_p, bias, d_rpe, apply_q_rpe)
            self.norm_src = nn.LayerNorm(d_model)
        self.attn = AttentionRPE(d_model, n_head, dropout_p, bias, d_rpe, apply_q_rpe)
        self.linear1 = nn.Linear(d_model, k_feedforward * d_model)
        self.linear2 = nn.Linear(k_feedforward * d_model, d_model)
        self.norm2 = nn.LayerNorm(d_model)


class TransformerBlockRPE(nn.Module):
    def __init__(self, d_model: int, n_head: int = 4, k_feedforward: int = 4, dropout_p: float = 0.1, bias: bool = True,
                 activation: str = "relu", out_layernorm: bool = False, apply_q_rpe: bool = False, n_layer: int = 1,
                 mode: str = "enc_self_attn", d_rpe: int = -1) -> None:
        super().__init__()
        assert mode in ("enc_self_attn", "enc_cross_attn", "dec_cross_attn")
        if out_layernorm:
            raise NotImplementedError("out_layernorm is off in the default config")
        self.mode, self.dropout_p, self.out_layernorm = mode, dropout_p, None
        self.layers = nn.ModuleList([
            TransformerRPE(d_model, n_head, k_feedforward, dropout_p, bias, activation, mode, d_rpe, apply_q_rpe)
            for _ in range(n_layer)])

    def _forward_train(self, src, src_padding_mask, tgt, tgt_padding_mask, rpe, decoder_tgt, decoder_tgt_padding_mask, decoder_rpe):
        """train(): the differentiable path of training (train_graph.transformer_block: HIP attention forward / backward with keyed
        probability dropout, keyed residual / FFN dropouts, autograd through the projections) - transformer_rpe.py:207-245."""
        from ... import train_graph as TG

        n, S, d = src.shape
        dev = src.device
        x = src.reshape(n * S, d).contiguous().float()
        inv = src_padding_mask if src_padding_mask is not None else torch.zeros(n, S, dtype=torch.bool, device=dev)
        u8 = lambda m: m.to(torch.uint8).contiguous()
        if self.mode == "enc_self_attn":
            knn = dict(idx=tgt.to(torch.int32).contiguous(), invalid=u8(tgt_padding_mask), emb=rpe.float().contiguous())
            cross = None
        elif self.mode == "dec_cross_attn":
            knn = dict(idx=decoder_tgt.to(torch.int32).contiguous(), invalid=u8(decoder_tgt_padding_mask), emb=decoder_rpe.float().contiguous())
            K = tgt.shape[2]
            tokens = tgt.reshape(n * S * K, d).contiguous().float()  # the gathered cross targets as a table of S * K rows per entry
            idx = torch.arange(S * K, dtype=torch.int32, device=dev).view(1, S, K).expand(n, -1, -1).contiguous()
            m, e = u8(tgt_padding_mask), rpe.float().contiguous()
            cross = lambda layer: [TG.Targets(tokens, idx, m, e, S * K)]
        else:
            raise NotImplementedError("enc_cross_attn is only used by the non-default RNN variant")
        with TG.module_scope(n, dev):
            y = TG.transformer_block(self, x, inv, n, S, knn, cross, p=self.dropout_p, training=True)
        return y.view(n, S, d), None

    def forward(self, src: Tensor, src_padding_mask: Optional[Tensor] = None, tgt: Optional[Tensor] = None,
                tgt_padding_mask: Optional[Tensor] = None, rpe: Optional[Tensor] = None, decoder_tgt: Optional[Tensor] = None,
                decoder_tgt_padding_mask: Optional[Tensor] = None, decoder_rpe: Optional[Tensor] = None,
                attn_mask: Optional[Tensor] = None, need_weights: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference signature (transformer_rpe.py:48-81). `tgt` (enc_self_attn) / `decoder_tgt` (dec_cross_attn) are
        integer KNN indices [n,S,K]; the cross targets of dec_cross_attn are gathered features [n,S,K,d]."""
        if attn_mask is not None or need_weights:
            raise NotImplementedError
        if self.training:
            return self._forward_train(src, src_padding_mask, tgt, tgt_padding_mask, rpe, decoder_tgt, decoder_tgt_padding_mask, decoder_rpe)
        n, S, d = src.shape
        x = src.reshape(n * S, d).contiguous().float().clone()
        inv = src_padding_mask if src_padding_mask is not None else torch.zeros(n, S, dtype=torch.bool, device=src.device)
        if self.mode == "enc_self_attn":
            knn = SelfKnn(tgt.to(torch.int32), tgt_padding_mask, emb=rpe.float())
            run_block(self, x, inv, n, S, knn)
        elif self.mode == "dec_cross_attn":
            knn = SelfKnn(decoder_tgt.to(torch.int32), decoder_tgt_padding_mask, emb=decoder_rpe.float())
            K = tgt.shape[2]
            kv = kv_tables(tgt.reshape(n * S * K, d).contiguous().float(), [(l.norm_tgt, l.attn) for l in self.layers])
            idx = torch.arange(S * K, dtype=torch.int32, device=src.device).view(1, S, K).expand(n, -1, -1).contiguous()
            m = tgt_padding_mask.to(torch.uint8).contiguous()
            e = rpe.contiguous().float()
            run_block(self, x, inv, n, S, knn, cross=lambda l: [Seg(kv, l * 2 * D, l * 2 * D + D, S * K, idx, m, e)])
        else:
            raise NotImplementedError("enc_cross_attn is only used by the non-default RNN variant")
        return x.view(n, S, d), None
