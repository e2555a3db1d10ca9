"""Differentiable building blocks of the training schedule: the autograd Functions over the HIP kernels (tall LINEAR + weight gradient,
KNARPE attention forward / backward, LayerNorm, keyed dropout, the one-pass residual / relu glue, PointNet tail, masked max-pool, the
navigation predictor's pair layer, the per-step state machine) and the small helpers built on them (linear, layer_norm, residual,
relu_drop, mlp, pointnet, attention with folded weights, K/V tables). State (arithmetic class, dropout scope, per-step caches):
train_state.py. Assembled into encoders / rollout / loss by train_graph.py."""
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import hip
from . import train_state as ST
from .hip import Seg
from .train_state import (ATTN_MFMA_MIN_ROWS, HEADS_TILE, LN_BWD, LN_FWD, TALL_LINEAR, WGRAD_MIN_ROWS, _DropScope, _POLICY_SITE0, bf16_contractions,
                          module_scope, precision)

D, NH, DH = 128, 4, 32
ATTN_DBIAS_K = os.environ.get("TBX_ATTN_DBIAS_K", "0") == "1"  # (1: accumulate the (identically zero) gradient of rpe_k_bias as rounds 1-5 did)
ATTN_FOLD_KERNEL = os.environ.get("TBX_ATTN_FOLD_KERNEL", "1") != "0"  # (0: the torch algebra, for A/B runs)


# tbx_tall_linear takes 64-wide layers too (the PointNet LINEARs 128 -> 64 and their input gradients, on the zero-padded image); inside the
# training step they stay on the library's fp32 GEMM by default: measured 110.2 -> 110.5 scenes/s with them on the kernel, but under the bf16
# class the PointNets' max-pools then sit behind bf16 products and the step's agreement with the reference loosens 3-5 x (train_c2.npz: loss
# 2.0e-4 -> 1.4e-3, spot gradients 2.5e-2 -> 0.11 of the block's largest entry; the 16-scene recombination 3.4e-4 -> 9.6e-3 on the gradient
# norms), and the fp32 class picks up the split products' 3e-5 on small gradient entries. TBX_TALL_64=1 turns it on.
TALL_64 = os.environ.get("TBX_TALL_64", "0") == "1"


def _tall_ok(x: Tensor, k: int, n: int) -> bool:
    return TALL_LINEAR and hip.tall_linear_ok(x, k, n) and (TALL_64 or (k % 128 == 0 and n % 128 == 0))


class TallLinearFn(torch.autograd.Function):
    """F.linear over very many rows (the time-batched pass: [n_scene * T * tokens (* window), k]). Forward and input gradient
    are library GEMMs in the form the library is fast at (row-major activations x K-contiguous weights: 60-100 TF/s fp32
    measured; the input gradient therefore multiplies by an explicit W^T copy instead of the library's NN kernel), the weight /
    bias gradient - a reduction over 10^5..10^6 rows into a [n, k] block - is tbx_linear_wgrad (csrc/wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, want16=False):
        ctx.save_for_backward(x, w)
        # (the bfloat16 copy takes no gradient: without this the engine fills a [rows, n] bfloat16 tensor of zeros for it at every backward -
        # 36 fills / 1.1 GB per step, tools/train_aten_sources.py)
        ctx.set_materialize_grads(False)
        ctx.has_b = b is not None
        ctx.bf16 = bf16_contractions()  # (the backward runs after training_step has returned: it keeps the forward's class)
        if _tall_ok(x, w.shape[1], w.shape[0]):
            # K, N multiples of 128: tbx_tall_linear (split-bf16 matrix path, byte-bound: ~3x the library's exact-fp32 rate; one
            # product under the bf16 class)
            if want16:  # a K/V table: the rows as bfloat16 as well, written by the same launch (for the matrix-core attention forward)
                y16 = torch.empty(*x.shape[:-1], w.shape[0], dtype=torch.bfloat16, device=x.device)
                y = hip.tall_linear(x, w, b, bf16=ctx.bf16, out16=y16.view(-1, w.shape[0]))
                ctx.mark_non_differentiable(y16)
                return y, y16
            return hip.tall_linear(x, w, b, bf16=ctx.bf16)
        y = F.linear(x, w, b)
        if want16:
            y16 = y.to(torch.bfloat16)
            ctx.mark_non_differentiable(y16)
            return y, y16
        return y

    @staticmethod
    def backward(ctx, dy, _d16=None):
        if dy is None:
            return None, None, None, None
        x, w = ctx.saved_tensors
        dx, dw, db = _tall_linear_backward(x, w, dy, ctx.has_b, ctx.bf16, ctx.needs_input_grad)
        return dx, dw, db, None


def _tall_linear_backward(x, w, dy, has_b: bool, bf16: bool, needs):
    """(dx, dw, db) of y = x W^T + b over very many rows: dx on tbx_tall_linear with the W^T image, dW / db on tbx_linear_wgrad."""
    dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
    dx = None
    if needs[0]:
        if _tall_ok(dy, w.shape[0], w.shape[1]):
            dx = hip.tall_linear(dy, w, None, wt=True, bf16=bf16)
        else:
            dx = F.linear(dy, w.t().contiguous())
    dw = db = None
    if needs[1] or (has_b and needs[2]):
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        n, k = dy2.shape[1], x2.shape[1]
        if not hip.linear_wgrad_ok(dy2, x2):
            # odd widths (heads with 1 / 2 / 5 outputs, the 31- / 121-wide map MLP): zero-padded copies with 4-float rows
            # (the library's GEMM for [n, rows] x [rows, k] with n = 1 took 56 ms at 10^6 rows)
            dy2 = F.pad(dy2, (0, -n % 4))
            x2 = F.pad(x2, (0, -k % 4)) if (k % 4 or not x2.is_contiguous()) else x2
            if x2.data_ptr() % 16:  # a contiguous view at an odd offset: the kernel reads float4 rows
                x2 = x2.clone()
            assert hip.linear_wgrad_ok(dy2, x2)
        dw, db = hip.linear_wgrad(dy2, x2, has_b, bf16=bf16)
        dw, db = dw[:n, :k], (db[:n] if db is not None else None)
    return dx, dw, db


class TallLinearReluDropFn(torch.autograd.Function):
    """h = dropout(relu(x W^T + b)) as ONE launch (tbx_tall_linear_relu_drop: the FFN's linear1 / an MLP layer over the time-batched
    rows) instead of TallLinearFn + ReluDropFn - the pre-activation is never written or read back. Backward: relu' and the mask are read
    off h (tbx_relu_drop_bwd), then TallLinearFn's products. Bit-identical to the two-launch form (tests/test_hip_training.py)."""

    @staticmethod
    def forward(ctx, x, w, b, p, seed, site, rows_per_scene, tb, t0):
        ctx.bf16 = bf16_contractions()
        h = hip.tall_linear(x, w, b, relu=True, bf16=ctx.bf16, drop=(p, seed, site, rows_per_scene, tb, t0))
        ctx.save_for_backward(x, w, h)
        ctx.has_b, ctx.p = b is not None, p
        return h

    @staticmethod
    def backward(ctx, dh):
        x, w, h = ctx.saved_tensors
        dz = hip.relu_drop_bwd(dh.contiguous(), h, ctx.p)
        dx, dw, db = _tall_linear_backward(x, w, dz, ctx.has_b, ctx.bf16, ctx.needs_input_grad)
        return dx, dw, db, None, None, None, None, None, None


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """F.linear; over >= WGRAD_MIN_ROWS rows with gradients on: TallLinearFn."""
    if torch.is_grad_enabled() and x.is_cuda and x.numel() // max(x.shape[-1], 1) >= WGRAD_MIN_ROWS and (w.requires_grad or x.requires_grad):
        return TallLinearFn.apply(x, w, b)
    return F.linear(x, w, b)


def linear_kv(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """`linear` for a K/V table: under the bf16 class the rows are also kept as bfloat16 for the matrix-core attention forward."""
    if (ST._KV16 is not None and bf16_contractions() and torch.is_grad_enabled() and x.is_cuda
            and x.numel() // max(x.shape[-1], 1) >= WGRAD_MIN_ROWS and (w.requires_grad or x.requires_grad)):
        y, y16 = TallLinearFn.apply(x, w, b, True)
        ST._KV16[y.data_ptr()] = (y, y16)
        return y
    return linear(x, w, b)


# ------------------------------------------------------------------------------------------------ attention
class KnarpeAttnFn(torch.autograd.Function):
    """out [rows, 640] = [sum_t a v | sum_t a e (4 heads)], flag [rows] (no valid target) for 1-2 target segments."""

    @staticmethod
    def _segs(kvs, meta):
        # meta per segment: (idx, invalid, emb | None, rel | None, n_tgt, batch_div[, (inv_ptr, inv_list) | None])
        return [Seg(kv, 0, D, m[4], m[0], m[1], m[2], m[5], rel=m[3]) for kv, m in zip(kvs, meta)]

    @staticmethod
    def forward(ctx, qbuf, bias_k, n, S, meta, freqs, drop, *kvs):
        # qbuf [rows, 640] = q | qt (4 heads x 128); kvs: K|V tables [tokens, 256];
        # freqs = (pose_rpe.pe_xy.freqs, pose_rpe.pe_yaw.freqs) or (None, None);
        # drop = None or (p, seed tensor, call id): dropout on the attention probabilities (attention_rpe.py:171-172)
        qbuf = qbuf.contiguous()
        kvs = [kv.contiguous() for kv in kvs]
        out = torch.empty(n * S, D + NH * D, dtype=torch.float32, device=qbuf.device)
        flag = torch.empty(n * S, dtype=torch.uint8, device=qbuf.device)
        bias_k = bias_k.contiguous()
        segs = KnarpeAttnFn._segs(kvs, meta)
        mfma = False
        if bf16_contractions() and n * S >= ATTN_MFMA_MIN_ROWS and freqs[0] is not None and all(sg.rel is not None and sg.emb is None for sg in segs):
            from . import engine

            mfma = engine.mfma_attention_ok(qbuf, 0, D, segs, out)
        if mfma:  # bf16 operands on the matrix cores, the VALU kernels' dropout mask; the backward below is the fp32 one either way
            if ST._KV16 is not None:  # tables that exist as bfloat16 (written by their producing LINEAR): half the gathered bytes
                k16 = [ST._KV16.get(kv.data_ptr()) for kv in kvs]
                if all(e is not None and e[0] is kv or (e is not None and e[0].data_ptr() == kv.data_ptr() and e[0].shape == kv.shape) for e, kv in zip(k16, kvs)):
                    segs = KnarpeAttnFn._segs([e[1] for e in k16], meta)
            hip.knarpe_attn_mfma(qbuf, 0, D, n, S, segs, out, flag, *freqs, drop=drop)
        else:
            hip.knarpe_attn(qbuf, 0, D, bias_k, n, S, segs, out, flag, *freqs, drop=drop)
        ctx.save_for_backward(qbuf, bias_k, *kvs)
        ctx.meta, ctx.n, ctx.S, ctx.freqs, ctx.drop = meta, n, S, freqs, drop
        ctx.mark_non_differentiable(flag)
        ctx.set_materialize_grads(False)  # (no zero fill for the flag's gradient at every backward)
        return out, flag

    @staticmethod
    def backward(ctx, dout, _dflag):
        if dout is None:
            return (None,) * (7 + len(ctx.saved_tensors) - 2)
        qbuf, bias_k, *kvs = ctx.saved_tensors
        meta, n, S = ctx.meta, ctx.n, ctx.S
        dq = torch.empty_like(qbuf)
        # d(bias_k) is identically zero in exact arithmetic (q_h . bk_h shifts every score of a row; the softmax ignores it): the kernels'
        # per-row accumulation [rows, 128] + its column sum returned round-off. ATTN_DBIAS_K = True restores that (the op-level test
        # checks the kernel's figure against autograd's, which is the same kind of round-off).
        db = torch.empty(qbuf.shape[0], D, dtype=bias_k.dtype, device=bias_k.device) if ATTN_DBIAS_K else None
        inv = [m[6] if len(m) > 6 else None for m in meta]
        gather = all(i is not None for i in inv)
        # gather mode overwrites every K|V row (the tables here are exactly [tokens, 256] = K|V); the atomics path accumulates
        dkv = [torch.empty_like(kv) if gather and kv.shape[1] == 2 * D else torch.zeros_like(kv) for kv in kvs]
        if gather:  # inverse K-nearest lists: dK / dV gathered per target token, no atomics
            hip.knarpe_attn_bwd_gather(qbuf, 0, D, bias_k, n, S, KnarpeAttnFn._segs(kvs, meta), dout.contiguous(), dq, dkv, db, inv,
                                       *ctx.freqs, drop=ctx.drop)
        else:
            hip.knarpe_attn_bwd(qbuf, 0, D, bias_k, n, S, KnarpeAttnFn._segs(kvs, meta), dout.contiguous(), dq, dkv, db, *ctx.freqs,
                                drop=ctx.drop)
        return (dq, db.sum(0) if db is not None else torch.zeros_like(bias_k), None, None, None, None, None, *dkv)


class Targets:
    """One target segment in table form: tokens [n_tables*T, 128] (already normalised), KNN set, sharing factor."""

    def __init__(self, tokens: Tensor, idx: Tensor, invalid: Tensor, emb: Optional[Tensor], n_tgt: int, batch_div: int = 1,
                 cache: Optional[dict] = None, key: Optional[str] = None, rel: Optional[Tensor] = None, freqs=(None, None),
                 inv=None):
        """Pose information per pair: `emb` [n,S,K,128] materialised, or `rel` [n,S,K,3] + freqs (rebuilt in-kernel)."""
        self.tokens, self.idx, self.invalid, self.emb, self.n_tgt, self.batch_div = tokens, idx, invalid, emb, n_tgt, batch_div
        self.rel, self.freqs, self.inv = rel, freqs, inv
        self.cache, self.key = cache, key  # static targets (map tokens): K/V tables computed once per training step


class LayerNormFn(torch.autograd.Function):
    """F.layer_norm over rows of 128 as tbx_layernorm_fwd (the row chains' arithmetic) / tbx_layernorm_bwd (x and dy read once, dx
    written once, deterministic dgamma / dbeta) instead of aten's four kernels."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        if LN_FWD:
            y, mean, rstd = hip.layernorm_fwd(x, w, b, eps)
        else:  # aten's forward (it hands over the per-row mean / rstd too)
            y, mean, rstd = torch.native_layer_norm(x, (x.shape[-1],), w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = hip.layernorm_bwd(x, dy.contiguous(), w, mean, rstd)
        return dx, dw, db, None


class LayerNormJoinFn(torch.autograd.Function):
    """(x, LayerNorm(x)) for the pre-norm residual form x' = x + f(LayerNorm(x)) (transformer_rpe.py:207-245): x is handed back so that
    BOTH of its gradients - the residual branch's and the LayerNorm's - arrive at this node, and the backward is ONE pass
    dx = d_residual + LayerNorm'(dy) (tbx_layernorm_bwd_add) where autograd ran tbx_layernorm_bwd and then summed the two with a kernel
    of its own: 92 such sums / 6.6 GB of traffic per 16-scene step (tools/train_aten_sources.py). Same values (a + b = b + a)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        assert x.is_contiguous()
        if LN_FWD:
            y, mean, rstd = hip.layernorm_fwd(x, w, b, eps)
        else:
            y, mean, rstd = torch.native_layer_norm(x, (x.shape[-1],), w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.set_materialize_grads(False)
        return x, y  # (an input returned as an output: autograd hands out a view of it whose grad_fn is this node)

    @staticmethod
    def backward(ctx, dres, dy):
        if dy is None:
            return dres, None, None, None
        x, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = hip.layernorm_bwd(x, dy.contiguous(), w, mean, rstd, add=None if dres is None else dres.contiguous())
        return dx, dw, db, None


LN_JOIN = os.environ.get("TBX_LN_JOIN", "1") != "0"  # (0: layer_norm + autograd's own sum, for A/B runs)


def layer_norm_join(x: Tensor, m):
    """-> (x', LayerNorm(x)): continue the residual branch from x' (LayerNormJoinFn); plain (x, layer_norm(x, m)) where that does not apply."""
    if (LN_JOIN and LN_BWD and m.weight.shape == (D,) and hip.layernorm_bwd_ok(x) and torch.is_grad_enabled() and x.requires_grad
            and x.is_contiguous()):
        return LayerNormJoinFn.apply(x, m.weight, m.bias, m.eps)
    return x, layer_norm(x, m)


def layer_norm(x: Tensor, m) -> Tensor:
    """m = an nn.LayerNorm over the last dimension."""
    if LN_BWD and m.weight.shape == (D,) and hip.layernorm_bwd_ok(x) and torch.is_grad_enabled():
        return LayerNormFn.apply(x, m.weight, m.bias, m.eps)
    return F.layer_norm(x, m.weight.shape, m.weight, m.bias, m.eps)


class KeyedDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, site, rows_per_scene, tb, t0):
        ctx.args = (p, seed, site, rows_per_scene, tb, t0)
        return hip.keyed_dropout(x.contiguous(), p, seed, site, rows_per_scene, tb, t0)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site, rows_per_scene, tb, t0 = ctx.args
        return hip.keyed_dropout(dy.contiguous(), p, seed, site, rows_per_scene, tb, t0), None, None, None, None, None, None


class ResidualDropFn(torch.autograd.Function):
    """zero_out[row] ? 0 : x + dropout(zero_y[row] ? 0 : y) in one pass (tbx_residual_drop_fwd / _bwd); the backward regenerates the mask."""

    @staticmethod
    def forward(ctx, x, y, zero_y, zero_out, p, seed, site, rows_per_scene, tb, t0):
        ctx.drop = (p, seed, site, rows_per_scene, tb, t0)
        ctx.save_for_backward(zero_y, zero_out)
        return hip.residual_drop_fwd(x.contiguous(), y.contiguous(), zero_y, zero_out, ctx.drop)

    @staticmethod
    def backward(ctx, dout):
        zero_y, zero_out = ctx.saved_tensors
        dy, dx = hip.residual_drop_bwd(dout.contiguous(), zero_y, zero_out, ctx.drop)
        return dx, dy, None, None, None, None, None, None, None, None


class ReluDropFn(torch.autograd.Function):
    """dropout(relu(z)) in one pass (tbx_relu_drop_fwd / _bwd); relu' and the mask are read off h > 0."""

    @staticmethod
    def forward(ctx, z, p, seed, site, rows_per_scene, tb, t0):
        h = hip.relu_drop_fwd(z.contiguous(), (p, seed, site, rows_per_scene, tb, t0))
        ctx.p = p
        ctx.save_for_backward(h)
        return h

    @staticmethod
    def backward(ctx, dh):
        (h,) = ctx.saved_tensors
        return hip.relu_drop_bwd(dh.contiguous(), h, ctx.p), None, None, None, None, None, None


def _glue_ok(x: Tensor, p: float, training: bool) -> bool:
    """The one-pass glue ops apply: device tensor, and a live dropout has its keyed scope (else torch's generator: the plain ops)."""
    return ST.GLUE_FUSED and hip.glue_ok(x) and not (training and p > 0 and ST._DROP is None)


def _drop_args(x: Tensor, p: float, training: bool):
    """tbx_keyed_dropout's arguments for x [..., cols] with the id _drop would give this site (advances it), or hip.NO_DROP."""
    if not (training and p > 0):
        return hip.NO_DROP
    ST._DROP["site"] += 1
    rows = x.numel() // x.shape[-1]
    assert rows % ST._DROP["n_batch"] == 0
    return (float(p), ST._DROP["seed"], ST._DROP["site"], rows // ST._DROP["n_batch"], ST._DROP["tb"], ST._DROP["t0"])


def residual(x: Tensor, y: Tensor, p: float, training: bool, zero_y: Optional[Tensor] = None, zero_out: Optional[Tensor] = None) -> Tensor:
    """(x + dropout(y.masked_fill(zero_y, 0))).masked_fill(zero_out, 0); zero_* u8 / bool per row ([rows]) or None."""
    if _glue_ok(x, p, training):
        u8 = lambda m: None if m is None else m.reshape(-1).to(torch.uint8).contiguous()
        return ResidualDropFn.apply(x, y, u8(zero_y), u8(zero_out), *_drop_args(y, p, training))
    if zero_y is not None:
        y = y.masked_fill(zero_y.reshape(-1).bool().unsqueeze(-1), 0.0)
    x = x + _drop(y, p, training)
    return x if zero_out is None else x.masked_fill(zero_out.reshape(-1).bool().unsqueeze(-1), 0.0)


def relu_drop(z: Tensor, p: float, training: bool) -> Tensor:
    if _glue_ok(z, p, training):
        return ReluDropFn.apply(z, *_drop_args(z, p, training))
    return _drop(F.relu(z), p, training)


LINEAR_RELU_DROP = os.environ.get("TBX_LINEAR_RELU_DROP", "1") != "0"  # (0: LINEAR and relu + dropout as two launches, for A/B runs)


def linear_relu_drop(x: Tensor, w: Tensor, b: Optional[Tensor], p: float, training: bool) -> Tensor:
    """dropout(relu(F.linear(x, w, b))): over >= WGRAD_MIN_ROWS rows of the time-batched pass one launch (TallLinearReluDropFn), else
    `relu_drop(linear(.))`. The dropout site id is taken exactly where relu_drop would take it."""
    rows = x.numel() // max(x.shape[-1], 1)
    if (LINEAR_RELU_DROP and TALL_LINEAR and torch.is_grad_enabled() and x.is_cuda and rows >= WGRAD_MIN_ROWS and (w.requires_grad or x.requires_grad)
            and x.dtype == torch.float32 and _tall_ok(x, w.shape[1], w.shape[0]) and ST.GLUE_FUSED
            and not (training and p > 0 and ST._DROP is None)):
        return TallLinearReluDropFn.apply(x, w, b, *_drop_args(x, p, training))
    return relu_drop(linear(x, w, b), p, training)


class AttnFoldFn(torch.autograd.Function):
    """The folded weights of one AttentionRPE module as ONE launch forward and ONE backward (tbx_attn_fold_fwd / _bwd) instead of the
    ~10 + ~25 slice / bmm / cat kernels torch ran per module: x 40 attention modules x 2 passes (no-grad stepping pass, differentiated
    pass) these were ~2,000 of a training step's ~7,000 launches, every one on 64 K floats at most."""

    @staticmethod
    def forward(ctx, w, b, wr, br, wo, bo):
        ctx.save_for_backward(w, b, wr, br, wo)
        return hip.attn_fold_fwd(w.contiguous(), b.contiguous(), wr.contiguous(), br.contiguous(), wo.contiguous(), bo.contiguous())

    @staticmethod
    def backward(ctx, *grads):
        w, b, wr, br, wo = ctx.saved_tensors
        return hip.attn_fold_bwd(w.contiguous(), b.contiguous(), wr.contiguous(), br.contiguous(), wo.contiguous(), grads)


def fold_attention_weights_torch(attn):
    """`fold_attention_weights` as torch algebra: what AttnFoldFn is tested against (tests/test_hip_training.py), and the form taken by
    modules that are not on the device."""
    W, b = attn.in_proj_weight, attn.in_proj_bias
    wr, br = attn.linear_rpe.weight, attn.linear_rpe.bias
    # the block-diagonal products head by head as batched GEMMs (B_k / B_v are never materialised)
    wq, wo = W[:D], attn.out_proj_weight
    wk_h = wr[:D].view(NH, DH, D)                                                       # B_k's blocks  [h][32, 128]
    wv_h = wr[D:].view(NH, DH, D)                                                       # B_v^T's blocks
    bk_wq = torch.bmm(wk_h.transpose(1, 2), wq.view(NH, DH, D)).reshape(NH * D, D)      # B_k^T W_q   [512, 128]
    bk_bq = torch.bmm(wk_h.transpose(1, 2), b[:D].view(NH, DH, 1)).reshape(NH * D)      # B_k^T b_q   [512]
    wo_bv = torch.bmm(wo.view(D, NH, DH).transpose(0, 1), wv_h).transpose(0, 1).reshape(D, NH * D)  # W_o B_v^T  [128, 512]
    return dict(w_in=torch.cat([wq, bk_wq], 0), b_in=torch.cat([b[:D], bk_bq], 0), w_kv=W[D:], b_kv=b[D:], bias_k=br[:D],
                w_out=torch.cat([wo, wo_bv], 1), b_out=wo @ br[D:] + attn.out_proj_bias)


def fold_attention_weights(attn):
    """The exact algebra of DESIGN.md §3 as GEMM weights:
      [q | qt] = x W_in^T + b_in        with  W_in  = [I | B_k]^T W_q           (640 x 128),  b_in  = [I | B_k]^T b_q
      y        = [sum a v | sum a e] W_out^T + b_out  with  W_out = W_o [I ; B_v]^T (128 x 640), b_out = W_o b_rpe_v + b_o
    B_k (128 x 512) / B_v (512 x 128): per-head blocks of linear_rpe's key / value halves. Also the K|V slice of in_proj."""
    ck = (id(attn), torch.is_grad_enabled())  # a no-grad pass must not hand its graph-less tensors to a differentiated one
    if ST._FOLD_CACHE is not None and ck in ST._FOLD_CACHE:
        return ST._FOLD_CACHE[ck]
    W = attn.in_proj_weight
    if W.is_cuda and tuple(W.shape) == (3 * D, D) and tuple(attn.linear_rpe.weight.shape) == (2 * D, D) and ATTN_FOLD_KERNEL:
        o = AttnFoldFn.apply(W, attn.in_proj_bias, attn.linear_rpe.weight, attn.linear_rpe.bias, attn.out_proj_weight, attn.out_proj_bias)
        f = dict(zip(("w_in", "b_in", "w_kv", "b_kv", "bias_k", "w_out", "b_out"), o))
    else:
        f = fold_attention_weights_torch(attn)
    if torch.is_grad_enabled() and W.is_cuda:
        # the module's three tall LINEARs (TallLinearFn: q | qt, K | V, out-projection) and their input gradients: six images, ONE launch at
        # the first request for any of them (none if the module's rows stay on the library's GEMM)
        from . import hip_base

        hip_base.pack_group([f["w_in"], f["w_kv"], f["w_out"]],
                            [(f[w], f[b], False, 1) for w, b in (("w_in", "b_in"), ("w_kv", "b_kv"), ("w_out", "b_out"))]
                            + [(f[w], None, True, 1) for w in ("w_in", "w_kv", "w_out")])
    if ST._FOLD_CACHE is not None:
        ST._FOLD_CACHE[ck] = f
    return f


def kv_table(attn, norm, t: Targets) -> Tensor:
    """K|V table [tokens, 256] of a target set for one attention layer (LayerNorm + projection, before the gather)."""
    f = fold_attention_weights(attn)
    make = lambda: linear_kv(layer_norm(t.tokens, norm) if norm is not None else t.tokens,
                               f["w_kv"], f["b_kv"])
    if t.cache is None or t.key is None:
        return make()
    k = (t.key, id(attn))
    if k not in t.cache:
        t.cache[k] = make()
    return t.cache[k]


def attention(attn, xq: Tensor, targets: Sequence[Targets], kvs: Sequence[Tensor], n: int, S: int, raw: bool = False):
    """attention_rpe.py:83-198 (rpe branch) in the factorised table form; xq [n*S, 128] is the normalised source.
    raw: -> (out-projection of every row, u8 flag of the rows without a valid target) for a caller that zeroes those rows itself."""
    f = fold_attention_weights(attn)
    qbuf = linear(xq, f["w_in"], f["b_in"])
    meta = [(t.idx, t.invalid, t.emb, t.rel, t.n_tgt, t.batch_div, t.inv) for t in targets]
    freqs = next((t.freqs for t in targets if t.rel is not None), (None, None))
    drop = None
    if ST._DROP is not None and attn.training and attn.dropout_p > 0:
        ST._DROP["call"] += 1
        assert n == ST._DROP["n_batch"], "attention call outside its dropout scope"
        drop = (float(attn.dropout_p), ST._DROP["seed"], ST._DROP["call"], ST._DROP["tb"], ST._DROP["t0"])
    out, flag = KnarpeAttnFn.apply(qbuf, f["bias_k"], n, S, meta, freqs, drop, *kvs)
    y = linear(out, f["w_out"], f["b_out"])
    if raw:
        return y, flag
    return y.masked_fill(flag.bool().unsqueeze(-1), 0.0)


def _drop(x: Tensor, p: float, training: bool) -> Tensor:
    """F.dropout of the reference as tbx_keyed_dropout (x [..., cols], batch entries = the scope's n_batch)."""
    if not (training and p > 0):
        return x
    if ST._DROP is None:  # outside a training step (unit tests of single modules): torch's generator
        return F.dropout(x, p, True)
    ST._DROP["site"] += 1
    rows = x.numel() // x.shape[-1]
    assert rows % ST._DROP["n_batch"] == 0
    return KeyedDropoutFn.apply(x, float(p), ST._DROP["seed"], ST._DROP["site"], rows // ST._DROP["n_batch"], ST._DROP["tb"], ST._DROP["t0"])


def _attn_residual(x: Tensor, y_flag, p: float, training: bool) -> Tensor:
    """x + dropout(y with the rows that had no valid target zeroed) (transformer_rpe.py:93-131 around attention_rpe.py:188-190)."""
    y, flag = y_flag
    return residual(x, y, p, training, zero_y=flag)


# ------------------------------------------------------------------------------------------------ small modules
def mlp(m, x: Tensor, training: bool = False) -> Tensor:
    """modules/mlp.py:69-72 (Linear [+LN] [+ReLU] [+Dropout] per layer)."""
    p = m.dropout_p
    for lin, lnm, act in m.linear_layers():
        if act and lnm is None and hip.glue_ok(x) and lin.weight.shape[0] % 4 == 0:
            x = linear_relu_drop(x, lin.weight, lin.bias, p, training)  # LINEAR + relu + dropout as ONE launch over the time-batched rows
            continue
        x = linear(x, lin.weight, lin.bias)
        if lnm is not None:
            x = layer_norm(x, lnm)
        if act and hip.glue_ok(x):
            x = relu_drop(x, p, training)  # relu + dropout as ONE launch forward and ONE backward (same site id as _drop's)
            continue
        if act:
            x = F.relu(x)
        x = _drop(x, p, training)
    return x


class PointNetTailFn(torch.autograd.Function):
    """[h | max over the group's valid rows of h] with invalid rows zeroed, h = dropout(relu(z)): tbx_pointnet_tail_fwd / _bwd."""

    @staticmethod
    def forward(ctx, z, inv8, p, seed, site, rows_per_scene, tb, t0):
        out = hip.pointnet_tail_fwd(z.contiguous(), inv8, None if p <= 0 else (p, seed, site, rows_per_scene, tb, t0))
        ctx.save_for_backward(out, inv8)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dout):
        out, inv8 = ctx.saved_tensors
        return hip.pointnet_tail_bwd(dout.contiguous(), out, inv8, ctx.p), None, None, None, None, None, None, None


class MaskedMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inv8):
        x = x.contiguous()
        ctx.save_for_backward(x, inv8)
        return hip.masked_maxpool_fwd(x, inv8)

    @staticmethod
    def backward(ctx, dy):
        x, inv8 = ctx.saved_tensors
        return hip.masked_maxpool_bwd(dy.contiguous(), x, inv8), None


def _pointnet_fused_ok(enc, x: Tensor, training: bool) -> bool:
    if not (ST.POINTNET_FUSED and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and 0 < x.shape[1] <= 32 and x.shape[0] > 0):
        return False
    for m in enc.mlp_layers:
        ll = m.linear_layers()
        if len(ll) != 1 or ll[0][1] is not None or not ll[0][2] or ll[0][0].weight.shape[0] != 64:
            return False
        if training and m.dropout_p > 0 and ST._DROP is None:  # torch's generator (unit tests of single modules): the plain ops
            return False
    return True


def pointnet(enc, x: Tensor, invalid: Tensor, training: bool = False) -> Tensor:
    """polyline_encoder.py:49-61 + pooling.py:18-19,38. x [G, W, 128], invalid [G, W] bool -> [G, 128]."""
    if _pointnet_fused_ok(enc, x, training):
        # per layer: the Linear, then ONE launch for relu / dropout / masked max / concat / zeroing (and one for their backward)
        inv8 = invalid.to(torch.uint8).contiguous()
        for m in enc.mlp_layers:
            lin = m.linear_layers()[0][0]
            z = linear(x, lin.weight, lin.bias)
            drop = (0.0, None, 0, 1, 1, 0)
            if training and m.dropout_p > 0:
                ST._DROP["site"] += 1  # the id _drop would have given this layer's dropout
                rows = z.shape[0] * z.shape[1]
                assert rows % ST._DROP["n_batch"] == 0
                drop = (float(m.dropout_p), ST._DROP["seed"], ST._DROP["site"], rows // ST._DROP["n_batch"], ST._DROP["tb"], ST._DROP["t0"])
            x = PointNetTailFn.apply(z, inv8, *drop)
        return MaskedMaxPoolFn.apply(x, inv8)
    im = invalid.unsqueeze(-1)
    for m in enc.mlp_layers:
        h = mlp(m, x, training).masked_fill(im, float("-inf"))
        x = torch.cat([h, h.amax(dim=1, keepdim=True).expand(-1, h.shape[1], -1)], -1).masked_fill(im, 0.0)
    y = x.masked_fill(im, float("-inf")).amax(1)
    return y.masked_fill(invalid.all(-1, keepdim=True), 0.0)


class NaviPairFirstLayer(torch.autograd.Function):
    """First Linear of NaviPredictor's pair MLP (navigation.py:245-262) without the [n, A, M, 384] concatenation:
        W [128, 384] = [W_a | W_m | W_e]:  h[n, a, m] = W_a f_a[n, a] + (W_m f_m[n, m] + b) + W_e e(rel[n, a, m])
    The per-agent and per-polyline terms are [n, A, 128] / [n, M, 128] GEMMs handed in; this function adds the per-pair term, with
    the 128-d pose embedding e rebuilt from the 12-byte relative pose scene by scene (tbx_pose_embed) in forward AND backward -
    neither the concatenation (1,536 B per pair) nor the embedding (512 B per pair) is kept for autograd: 12 B per pair are."""

    @staticmethod
    def forward(ctx, rel, w_e, pa, pm, fxy, fyw, relu=False):
        n, A, M, _ = rel.shape
        d = w_e.shape[0]
        w_c = w_e.contiguous()  # [128 out, 128 k], k-contiguous: the GEMM form the library is fast at (a strided slice of the
        # 384-wide weight sent it to a 1.2 TF/s kernel: 28 ms per training step)
        h = torch.empty(n, A, M, d, dtype=torch.float32, device=rel.device)
        ctx.bf16 = bf16_contractions()
        tall = TALL_LINEAR and A * M >= WGRAD_MIN_ROWS and d % 128 == 0 and w_c.shape[1] % 128 == 0
        for i in range(n):
            emb = hip.pose_embed(rel[i].reshape(-1, 3), fxy, fyw, w_c.shape[1])
            hi = h[i].view(A * M, d)
            if tall:  # (65 k rows x 128 x 128: the tall-LINEAR kernel of the step's arithmetic class)
                hip.tall_linear(emb, w_c, None, bf16=ctx.bf16, out=hi)
            else:
                torch.mm(emb, w_c.t(), out=hi)
        # + the per-agent and per-polyline terms (and the layer's relu, when it follows directly) in ONE pass over all scenes
        # (was: h[i] += pa[i][:, None] + pm[i][None] per scene - add, add_, copy_: 48 launches, 3 GB of traffic - and F.relu over the whole)
        ctx.relu = bool(relu) and hip.glue_ok(h)
        if hip.glue_ok(h):
            hip.pair_bias_relu(h, pa, pm, ctx.relu)
        else:
            h += pa.unsqueeze(2) + pm.unsqueeze(1)
        ctx.save_for_backward(rel, w_e, fxy, fyw, *((h,) if ctx.relu else ()))
        return h

    @staticmethod
    def backward(ctx, dh):
        rel, w_e, fxy, fyw, *hs = ctx.saved_tensors
        n, A, M, _ = rel.shape
        dh = dh.contiguous()
        if ctx.relu:
            dh = hip.relu_drop_bwd(dh, hs[0], 0.0)  # relu' read off the output
        dw = torch.zeros(w_e.shape, dtype=torch.float32, device=dh.device)
        for i in range(n):
            emb = hip.pose_embed(rel[i].reshape(-1, 3), fxy, fyw, w_e.shape[1])
            dw += hip.linear_wgrad(dh[i].view(A * M, -1), emb, want_db=False, bf16=ctx.bf16)[0]  # dY^T X over 65 k rows: tbx_linear_wgrad
        return None, dw, dh.sum(2), dh.sum(1), None, None, None


class TrainChain:
    """Device-resident state + buffers of tbx_train_chain (csrc/train_chain.hip): the per-step state machine of the training
    rollout - what `training_rollout` spells out in ~75 elementwise torch ops per step - for one batch. `step(s, mean)` advances
    one step (stepping pass), `run_all(mean)` all T steps from the initial state (differentiated pass, through TrainChainFn)."""

    def __init__(self, wm, b, tf_mask: Tensor, T: int) -> None:
        model, dyn, rc = wm.model, wm.dynamics, wm.hp.differentiable_reward
        gt_valid, gt_pose, gt_motion = b["gt/ag_valid"], b["gt/ag_pose"], b["gt/ag_motion"]
        ag_type, dest = b["ref/ag_type"], b["gt/ag_navi"]
        n, A, Tg = gt_valid.shape
        dev, W = gt_pose.device, model.temp_window_size
        self.n, self.A, self.T, self.W, self.dev = n, A, T, W, dev
        if getattr(dyn, "_max_act", None) is None or dyn._max_act.device != dev:
            dyn._max_act = torch.tensor([[a, y] for a, y in zip(dyn.max_acc, dyn.max_yaw_rate)], device=dev)
        u8, f32 = torch.uint8, torch.float32
        bi = torch.arange(n, device=dev).unsqueeze(1)
        d_type = b["map/type"][bi, dest]
        d_dir = b["map/dir"][bi, dest][..., :2].float()
        N = d_dir.shape[2]
        k = dict(gt_valid=gt_valid.to(u8), gt_pose=gt_pose.float(), gt_motion=gt_motion.float(), tf_mask=tf_mask.to(u8),
                 lim=(ag_type.unsqueeze(-1) * dyn._max_act).sum(2).float(), dest_pos=b["map/pos"][bi, dest][..., :2].float(),
                 dest_dir=d_dir / torch.norm(d_dir, dim=-1, keepdim=True), dest_invalid=(~b["map/valid"][bi, dest]).to(u8),
                 dest_thresh=50.0 * (1 - d_type[:, :, 4].float() * 0.8),
                 dest_kind=d_type[:, :, :4].any(-1).to(u8) + 2 * d_type[:, :, 4].to(u8), boundary=b["map/boundary"].float())
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=dev)
        k.update(valid=z(n, A, dt=u8), disabled=z(n, A, dt=u8), navi_valid=z(n, A, dt=u8), outside=z(n, A, dt=u8), reached=z(n, A, dt=u8),
                 pose=z(n, A, 3), motion=z(n, A, 3), rec_valid=z(n, T + W, A, dt=u8), rec_pose=z(n, T + W, A, 3), rec_motion=z(n, T + W, A, 3),
                 rec_navi_valid=z(n, T + W, A, dt=u8), pred_valid=z(n, T, A, dt=u8), tf=z(n, T, A, dt=u8), ov=z(n, T, A, dt=u8),
                 reward_valid=z(n, T, A, dt=u8), pred_pose=z(n, T, A, 3), pred_motion=z(n, T, A, 3), reward=z(n, T, A))
        self.t = {name: v.contiguous() for name, v in k.items()}
        a = hip.TrainChainArgs()
        a.n_batch, a.n_ag, a.n_step, a.n_step_gt, a.n_node, a.window = n, A, T, Tg, N, W
        a.dt, a.w_pos, a.w_rot, a.w_spd = float(dyn.dt), float(rc.l_pos.weight), float(rc.l_rot.weight), float(rc.l_spd.weight)
        for name, v in self.t.items():
            setattr(a, name, v.data_ptr())
        self.args = a
        self.init = (gt_valid[:, :, 0].to(u8), gt_pose[:, :, 0].float(), gt_motion[:, :, 0].float(), gt_valid.any(-1).to(u8))
        # the policy inputs of the NEXT step, written by the step's own launch (tbx_train_chain_fwd_windows): windows + current flags
        self.win = dict(hv=z(n, A, W, dt=u8), hp=z(n, A, W, 3), hm=z(n, A, W, 3), valid=torch.zeros(n, A, dtype=torch.bool, device=dev),
                        navi_valid=torch.zeros(n, A, dtype=torch.bool, device=dev))
        self._win_step = 0  # the step whose inputs self.win holds (0: none)
        self.reset()

    def reset(self) -> None:
        """Initial state (Dynamics.init, dynamics.py:29-64) into the state buffers and into record slot W - 1."""
        t, W = self.t, self.W
        v, p, m, nv = self.init
        t["valid"].copy_(v), t["pose"].copy_(p), t["motion"].copy_(m), t["navi_valid"].copy_(nv)
        for name in ("disabled", "outside", "reached"):
            t[name].zero_()
        t["rec_valid"][:, W - 1].copy_(v), t["rec_pose"][:, W - 1].copy_(p), t["rec_motion"][:, W - 1].copy_(m)
        t["rec_navi_valid"][:, W - 1].copy_(nv)
        self._emit_windows(None, 0, 0)

    def _emit_windows(self, mean, t0: int, t1: int) -> None:
        w = self.win
        hip.train_chain_fwd_windows(self.args, mean, self.A * 2, 0, t0, t1, w["hv"], w["hp"], w["hm"], w["valid"], w["navi_valid"])
        self._win_step = t1 + 1

    def before(self, s: int):
        """Policy inputs of step s (1-based): windows (valid u8 [n,A,W], pose, motion [n,A,W,3]; oldest first) and the current
        (valid bool [n,A], pose [n,A,3], navi_valid bool [n,A])."""
        t, W = self.t, self.W
        if self._win_step == s:  # written by the launch that ran step s - 1 (or by reset): no copies
            w = self.win
            return w["hv"], w["hp"], w["hm"], w["valid"], t["rec_pose"][:, s - 1 + W - 1], w["navi_valid"]
        sl = slice(s - 1, s - 1 + W)
        hv = t["rec_valid"][:, sl].permute(0, 2, 1).contiguous()
        hp = t["rec_pose"][:, sl].permute(0, 2, 1, 3).contiguous()
        hm = t["rec_motion"][:, sl].permute(0, 2, 1, 3).contiguous()
        cur = s - 1 + W - 1
        return hv, hp, hm, t["rec_valid"][:, cur].bool(), t["rec_pose"][:, cur], t["rec_navi_valid"][:, cur].bool()

    def step(self, s: int, mean: Tensor) -> None:
        mean = mean.detach().reshape(self.n, self.A, 2).contiguous()
        self._emit_windows(mean, s - 1, s)  # the step + the next step's policy inputs in one launch

    def windows(self):
        """All steps' policy inputs in [scene][step] order: (hv [n*T,A,W] u8, hp, hm [n*T,A,W,3], valid [n*T,A] bool, pose
        [n*T,A,3], navi_valid [n*T,A] bool) from the records of a finished stepping pass."""
        t, W, T, n, A = self.t, self.W, self.T, self.n, self.A
        win = lambda x: x[:, :T + W - 1].unfold(1, W, 1)  # [n, T, A(,3), W]
        hv = win(t["rec_valid"]).reshape(n * T, A, W).contiguous()
        hp = win(t["rec_pose"]).permute(0, 1, 2, 4, 3).reshape(n * T, A, W, 3).contiguous()
        hm = win(t["rec_motion"]).permute(0, 1, 2, 4, 3).reshape(n * T, A, W, 3).contiguous()
        cur = slice(W - 1, W - 1 + T)
        return (hv, hp, hm, t["rec_valid"][:, cur].reshape(n * T, A).bool(), t["rec_pose"][:, cur].reshape(n * T, A, 3).contiguous(),
                t["rec_navi_valid"][:, cur].reshape(n * T, A).bool())

    def outputs(self, reward: Tensor) -> Dict[str, Tensor]:
        """The rollout log in training_rollout's layout ([n, A, T(,3)])."""
        t = self.t
        p = lambda x: x.permute(0, 2, 1) if x.dim() == 3 else x.permute(0, 2, 1, 3)
        return {"pred_valid": p(t["pred_valid"]).bool(), "pred_pose": p(t["pred_pose"]), "pred_motion": p(t["pred_motion"]),
                "reward": p(reward), "reward_valid": p(t["reward_valid"]).bool(), "tf": p(t["tf"]).bool()}


class TrainChainFn(torch.autograd.Function):
    """reward [n,T,A] of the whole rollout from the batched action means [n,T,A,2] (tbx_train_chain_fwd over all steps from the
    initial state); backward = tbx_train_chain_bwd (reverse walk over the steps)."""

    @staticmethod
    def forward(ctx, mean, chain):
        mean = mean.contiguous()
        chain.reset()
        hip.train_chain_fwd(chain.args, mean, chain.T * chain.A * 2, chain.A * 2, 0, chain.T)
        ctx.chain = chain
        ctx.save_for_backward(mean)
        return chain.t["reward"].clone()

    @staticmethod
    def backward(ctx, d_reward):
        (mean,) = ctx.saved_tensors
        chain = ctx.chain
        d_mean = torch.empty_like(mean)
        hip.train_chain_bwd(chain.args, mean, chain.T * chain.A * 2, chain.A * 2, d_reward.contiguous(), d_mean)
        return d_mean, None
