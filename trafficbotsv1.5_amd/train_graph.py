"""Differentiable (training) schedule of the hot path on the GPU.

Forward AND backward of the KNARPE attention are hand-written HIP kernels (tbx_knarpe_attn_fwd / _bwd) behind
`KnarpeAttnFn`; K-nearest selection, pose embeddings and feature preparation are the same HIP kernels as in inference
(no gradient flows through them: the reference computes them under no_grad, utils/rpe.py:7,40,61). The dense
projections are plain library GEMMs (`F.linear` -> hipBLASLt) and the LayerNorm / ReLU / masking / max-pool glue is
elementwise torch on the device, so autograd provides their backward. Fusing those into chain-backward kernels is the
next step; the formulation (K/V projected before the gather, linear_rpe folded) is identical to the
inference engine, so both are checked against the same oracle.

Everything takes the reference-named nn.Modules as parameter containers (same state dict as inference).

Time-batched rollout (`training_rollout_batched`, the default): the reference detaches the policy inputs of every
closed-loop step (waymo_motion.py:206-311 with training=True), so the only gradient path across steps is the dynamics
chain. The rollout therefore runs twice: (1) step by step WITHOUT autograd, recording each step's (detached) policy inputs;
(2) the 90 policy evaluations of a scene as ONE batch of 90 x n_scene entries WITH autograd (map K/V tables shared through
`batch_div`, dropout masks keyed by (site, step, row, column) so that the batched pass draws the masks of the sequential
one), followed by the tiny per-step dynamics / reward chain on the batched means. Same loss and gradients as stepping with
autograd (`training_rollout`, kept as the checked reference of the restructure), but ~25 large launches per layer
instead of 90 x as many small ones.
"""
import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.distributions import Categorical, Independent, Normal, kl_divergence

from . import hip
from .hip import Seg
from .models.modules.distributions import DestCategorical, DiagGaussian

D, NH, DH = 128, 4, 32


# ------------------------------------------------------------------------------------------------ arithmetic class
# Precision of the training step's CONTRACTIONS - what torch.autocast would switch (the reference trains at `precision: 16`,
# configs/trainer/default.yaml:16):
#   "bf16"  ONE bf16 product per term, fp32 accumulation: F.linear forward / input gradient over >= WGRAD_MIN_ROWS rows
#           (tbx_tall_linear_bf16), their weight gradients (tbx_linear_wgrad_bf16), the attention forward of the differentiated launches
#           (>= 193 source rows; tbx_knarpe_attn_fwd_mfma_dropout_tb: bf16 q, qt, K, V, e and softmax weights on the matrix cores). LayerNorm,
#           softmax, the attention backward (it recomputes the probabilities in fp32 and regenerates the forward's dropout mask), the
#           elementwise glue, the state machine, the losses and the optimizer stay fp32 - as under autocast. The default.
#   "fp32"  the fp32-class path of rounds 2-4 (split-bf16 products / exact-fp32 MFMA, VALU attention): the tight-tolerance parity path
#           (tests/test_hip_training.py runs both against the reference's golden loss / gradient norms).
# `wm.train_precision` overrides the default for one module; training_step makes it current for its forward (autograd Functions keep
# the flag they ran their forward with for their backward).
DEFAULT_PRECISION = os.environ.get("TBX_TRAIN_PRECISION", "bf16")
_PREC: Optional[str] = None


def precision() -> str:
    return _PREC or DEFAULT_PRECISION


def bf16_contractions() -> bool:
    return precision() == "bf16"


ATTN_MFMA_MIN_ROWS = 193  # (the differentiated launches of a training step are 10^4..10^5 rows; the stepping pass - 1024 rows - keeps the VALU ring kernel, see _agent_policy_engine)


# ------------------------------------------------------------------------------------------------ dense contractions
HEADS_TILE = os.environ.get("TBX_HEADS_TILE_TRAIN", "1") != "0"  # the stepping pass's heads as one tbx_heads_tile launch (raw inputs + keyed dropouts)
TALL_LINEAR = os.environ.get("TBX_TALL_LINEAR", "1") != "0"  # forward / input-gradient products of the time-batched pass on tbx_tall_linear
WGRAD_MIN_ROWS = 16384  # from here on dW = dY^T X is a reduction over so many rows that the library GEMM has 1-2 output tiles


class TallLinearFn(torch.autograd.Function):
    """F.linear over very many rows (the time-batched pass: [n_scene * T * tokens (* window), k]). Forward and input gradient
    are library GEMMs in the form the library is fast at (row-major activations x K-contiguous weights: 60-100 TF/s fp32
    measured; the input gradient therefore multiplies by an explicit W^T copy instead of the library's NN kernel), the weight /
    bias gradient - a reduction over 10^5..10^6 rows into a [n, k] block - is tbx_linear_wgrad (csrc/wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, want16=False):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.bf16 = bf16_contractions()  # (the backward runs after training_step has returned: it keeps the forward's class)
        if TALL_LINEAR and hip.tall_linear_ok(x, w.shape[1], w.shape[0]):
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
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            if TALL_LINEAR and hip.tall_linear_ok(dy, w.shape[0], w.shape[1]):
                dx = hip.tall_linear(dy, w, None, wt=True, bf16=ctx.bf16)
            else:
                dx = F.linear(dy, w.t().contiguous())
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
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
            dw, db = hip.linear_wgrad(dy2, x2, ctx.has_b, bf16=ctx.bf16)
            dw, db = dw[:n, :k], (db[:n] if db is not None else None)
        return dx, dw, db, None


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """F.linear; over >= WGRAD_MIN_ROWS rows with gradients on: TallLinearFn."""
    if torch.is_grad_enabled() and x.is_cuda and x.numel() // max(x.shape[-1], 1) >= WGRAD_MIN_ROWS and (w.requires_grad or x.requires_grad):
        return TallLinearFn.apply(x, w, b)
    return F.linear(x, w, b)


# K/V tables of the current training step that also exist as bfloat16 (made by the launch that made the fp32 rows: tbx_tall_linear_dual):
# fp32 table's data_ptr -> (the fp32 table - kept alive, so the address stays its own -, the bfloat16 copy). None outside a step.
_KV16: Optional[dict] = None


def linear_kv(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """`linear` for a K/V table: under the bf16 class the rows are also kept as bfloat16 for the matrix-core attention forward."""
    if (_KV16 is not None and bf16_contractions() and torch.is_grad_enabled() and x.is_cuda
            and x.numel() // max(x.shape[-1], 1) >= WGRAD_MIN_ROWS and (w.requires_grad or x.requires_grad)):
        y, y16 = TallLinearFn.apply(x, w, b, True)
        _KV16[y.data_ptr()] = (y, y16)
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
            if _KV16 is not None:  # tables that exist as bfloat16 (written by their producing LINEAR): half the gathered bytes
                k16 = [_KV16.get(kv.data_ptr()) for kv in kvs]
                if all(e is not None and e[0] is kv or (e is not None and e[0].data_ptr() == kv.data_ptr() and e[0].shape == kv.shape) for e, kv in zip(k16, kvs)):
                    segs = KnarpeAttnFn._segs([e[1] for e in k16], meta)
            hip.knarpe_attn_mfma(qbuf, 0, D, n, S, segs, out, flag, *freqs, drop=drop)
        else:
            hip.knarpe_attn(qbuf, 0, D, bias_k, n, S, segs, out, flag, *freqs, drop=drop)
        ctx.save_for_backward(qbuf, bias_k, *kvs)
        ctx.meta, ctx.n, ctx.S, ctx.freqs, ctx.drop = meta, n, S, freqs, drop
        ctx.mark_non_differentiable(flag)
        return out, flag

    @staticmethod
    def backward(ctx, dout, _dflag):
        qbuf, bias_k, *kvs = ctx.saved_tensors
        meta, n, S = ctx.meta, ctx.n, ctx.S
        dq = torch.empty_like(qbuf)
        db = torch.empty(qbuf.shape[0], D, dtype=bias_k.dtype, device=bias_k.device)  # per-row d(bias_k); summed below
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
        return (dq, db.sum(0), None, None, None, None, None, *dkv)


class Targets:
    """One target segment in table form: tokens [n_tables*T, 128] (already normalised), KNN set, sharing factor."""

    def __init__(self, tokens: Tensor, idx: Tensor, invalid: Tensor, emb: Optional[Tensor], n_tgt: int, batch_div: int = 1,
                 cache: Optional[dict] = None, key: Optional[str] = None, rel: Optional[Tensor] = None, freqs=(None, None),
                 inv=None):
        """Pose information per pair: `emb` [n,S,K,128] materialised, or `rel` [n,S,K,3] + freqs (rebuilt in-kernel)."""
        self.tokens, self.idx, self.invalid, self.emb, self.n_tgt, self.batch_div = tokens, idx, invalid, emb, n_tgt, batch_div
        self.rel, self.freqs, self.inv = rel, freqs, inv
        self.cache, self.key = cache, key  # static targets (map tokens): K/V tables computed once per training step


# Folded projection weights of the attention modules, valid while the parameters do not change: `training_step` opens a
# cache for its 90 closed-loop steps (the same weights serve every step), so the folding - and its backward - run once per
# training step and each attention call is [one GEMM -> tbx_knarpe_attn -> one GEMM]. None = no caching (fold per call).
_FOLD_CACHE: Optional[dict] = None
# Dropout of a training step: {"seed": int64[1] device tensor, "call": running attention call id, "site": running id of the
# elementwise dropout sites, "n_batch" / "tb" / "t0": batch entries of the current scope and its time batching (include/
# tbx_hip.h: entry b = step t0 + b % tb of scene b / tb)}. The seed lives on the device (a captured step draws new masks when
# the host rewrites it between replays); None = no dropout. Every mask is a hash of (seed, site | call, step, scene row, ...),
# so the time-batched pass of the rollout re-draws the masks of the step-by-step pass.
_DROP: Optional[dict] = None
_POLICY_SITE0 = 1 << 20  # site / call ids of a policy step restart here every step (the step number is part of the key)


class _DropScope:
    def __init__(self, n_batch: int, tb: int = 1, t0: int = 0, restart: Optional[int] = None):
        self.kw = dict(n_batch=n_batch, tb=tb, t0=t0)
        self.restart = restart

    def __enter__(self):
        if _DROP is not None:
            self.saved = dict(_DROP)
            _DROP.update(self.kw)
            if self.restart is not None:
                _DROP["site"] = _DROP["call"] = self.restart

    def __exit__(self, *exc):
        if _DROP is not None:
            site, call = _DROP["site"], _DROP["call"]
            _DROP.update(self.saved)
            if self.restart is None:  # ids keep running across the step's non-policy scopes
                _DROP["site"], _DROP["call"] = site, call
        return False


@contextlib.contextmanager
def module_scope(n_batch: int, device):
    """Dropout scope of ONE module call in train() outside a training step (the reference's modules run in train mode,
    modules/transformer_rpe.py:207-245, mlp.py:58-72): the keyed masks of this call hang off a seed drawn from torch's generator
    (torch.manual_seed reproduces them), elementwise sites and attention calls are numbered from 0. Inside a training step the
    step's own scope stays in force."""
    global _DROP
    if _DROP is not None:
        with _DropScope(n_batch):
            yield
        return
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    _DROP = {"seed": seed, "call": 0, "site": 0, "n_batch": n_batch, "tb": 1, "t0": 0}
    try:
        yield
    finally:
        _DROP = None


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


LN_BWD = os.environ.get("TBX_LN_BWD", "1") != "0"
LN_FWD = os.environ.get("TBX_LN_FWD", "1") != "0"


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


GLUE_FUSED = os.environ.get("TBX_GLUE_FUSED", "1") != "0"


def _glue_ok(x: Tensor, p: float, training: bool) -> bool:
    """The one-pass glue ops apply: device tensor, and a live dropout has its keyed scope (else torch's generator: the plain ops)."""
    return GLUE_FUSED and hip.glue_ok(x) and not (training and p > 0 and _DROP is None)


def _drop_args(x: Tensor, p: float, training: bool):
    """tbx_keyed_dropout's arguments for x [..., cols] with the id _drop would give this site (advances it), or hip.NO_DROP."""
    if not (training and p > 0):
        return hip.NO_DROP
    _DROP["site"] += 1
    rows = x.numel() // x.shape[-1]
    assert rows % _DROP["n_batch"] == 0
    return (float(p), _DROP["seed"], _DROP["site"], rows // _DROP["n_batch"], _DROP["tb"], _DROP["t0"])


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


def fold_attention_weights(attn):
    """The exact algebra of DESIGN.md §3 as GEMM weights:
      [q | qt] = x W_in^T + b_in        with  W_in  = [I | B_k]^T W_q           (640 x 128),  b_in  = [I | B_k]^T b_q
      y        = [sum a v | sum a e] W_out^T + b_out  with  W_out = W_o [I ; B_v]^T (128 x 640), b_out = W_o b_rpe_v + b_o
    B_k (128 x 512) / B_v (512 x 128): per-head blocks of linear_rpe's key / value halves. Also the K|V slice of in_proj."""
    ck = (id(attn), torch.is_grad_enabled())  # a no-grad pass must not hand its graph-less tensors to a differentiated one
    if _FOLD_CACHE is not None and ck in _FOLD_CACHE:
        return _FOLD_CACHE[ck]
    W, b = attn.in_proj_weight, attn.in_proj_bias
    wr, br = attn.linear_rpe.weight, attn.linear_rpe.bias
    # the block-diagonal products head by head as batched GEMMs (B_k / B_v are never materialised: building them with
    # torch.block_diag cost 8 slice copies per module forward and ~24 tiny kernels backward, x 40 attention modules per step)
    wq, wo = W[:D], attn.out_proj_weight
    wk_h = wr[:D].view(NH, DH, D)                                                       # B_k's blocks  [h][32, 128]
    wv_h = wr[D:].view(NH, DH, D)                                                       # B_v^T's blocks
    bk_wq = torch.bmm(wk_h.transpose(1, 2), wq.view(NH, DH, D)).reshape(NH * D, D)      # B_k^T W_q   [512, 128]
    bk_bq = torch.bmm(wk_h.transpose(1, 2), b[:D].view(NH, DH, 1)).reshape(NH * D)      # B_k^T b_q   [512]
    wo_bv = torch.bmm(wo.view(D, NH, DH).transpose(0, 1), wv_h).transpose(0, 1).reshape(D, NH * D)  # W_o B_v^T  [128, 512]
    f = dict(w_in=torch.cat([wq, bk_wq], 0), b_in=torch.cat([b[:D], bk_bq], 0), w_kv=W[D:], b_kv=b[D:], bias_k=br[:D],
             w_out=torch.cat([wo, wo_bv], 1), b_out=wo @ br[D:] + attn.out_proj_bias)
    if _FOLD_CACHE is not None:
        _FOLD_CACHE[ck] = f
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
    if _DROP is not None and attn.training and attn.dropout_p > 0:
        _DROP["call"] += 1
        assert n == _DROP["n_batch"], "attention call outside its dropout scope"
        drop = (float(attn.dropout_p), _DROP["seed"], _DROP["call"], _DROP["tb"], _DROP["t0"])
    out, flag = KnarpeAttnFn.apply(qbuf, f["bias_k"], n, S, meta, freqs, drop, *kvs)
    y = linear(out, f["w_out"], f["b_out"])
    if raw:
        return y, flag
    return y.masked_fill(flag.bool().unsqueeze(-1), 0.0)


# The stepping pass of the time-batched rollout needs no autograd: with gradients off, whole layers run as the inference
# engine's chain kernels (engine.run_block: ~6 launches per transformer layer instead of ~40 torch ops), with the keyed dropouts
# of training as DROPOUT stages / inside the attention kernels - same site / call ids, hence same masks, as the torch ops here.
NOGRAD_CHAINS = True


def _chains_ok(x: Tensor) -> bool:
    return NOGRAD_CHAINS and not torch.is_grad_enabled() and x.is_cuda and (_DROP is None or _DROP["tb"] == 1)


def _transformer_block_chains(block, x, src_invalid, n, S, self_knn, cross, p, training) -> Tensor:
    from .engine import SelfKnn, kv_tables, run_block

    layers = list(block.layers)
    xx = x.contiguous().clone()
    knn = SelfKnn(self_knn["idx"], self_knn["invalid"], emb=self_knn.get("emb"), rel=self_knn.get("rel"))
    cross_fn = None
    if block.mode == "dec_cross_attn":
        tg = list(cross(layers[0]))  # the same token sets for every layer
        tabs = []
        for t in tg:  # K|V tables of all layers in one chain launch ([tokens, n_layer * 256]); static sets: once per training step
            ck = None if (t.cache is None or t.key is None) else (t.key, id(block), "all-layers")
            if ck is None or ck not in t.cache:
                tab = kv_tables(t.tokens.contiguous(), [(l.norm_tgt, l.attn) for l in layers])
                if ck is not None:
                    t.cache[ck] = tab
            tabs.append(tab if ck is None else t.cache[ck])
        cross_fn = lambda l: [Seg(tab, l * 2 * D, l * 2 * D + D, t.n_tgt, t.idx, t.invalid, t.emb, t.batch_div, rel=t.rel)
                              for tab, t in zip(tabs, tg)]
    drop = None
    if training and _DROP is not None:
        drop = dict(p=float(p), seed=_DROP["seed"], site=_DROP["site"], call=_DROP["call"], step=_DROP["t0"])
    run_block(block, xx, src_invalid, n, S, knn, cross=cross_fn, freqs=self_knn.get("freqs"), drop=drop)
    if drop is not None:
        _DROP["site"], _DROP["call"] = drop["site"], drop["call"]
    return xx


def _drop(x: Tensor, p: float, training: bool) -> Tensor:
    """F.dropout of the reference as tbx_keyed_dropout (x [..., cols], batch entries = the scope's n_batch)."""
    if not (training and p > 0):
        return x
    if _DROP is None:  # outside a training step (unit tests of single modules): torch's generator
        return F.dropout(x, p, True)
    _DROP["site"] += 1
    rows = x.numel() // x.shape[-1]
    assert rows % _DROP["n_batch"] == 0
    return KeyedDropoutFn.apply(x, float(p), _DROP["seed"], _DROP["site"], rows // _DROP["n_batch"], _DROP["tb"], _DROP["t0"])


def _attn_residual(x: Tensor, y_flag, p: float, training: bool) -> Tensor:
    """x + dropout(y with the rows that had no valid target zeroed) (transformer_rpe.py:93-131 around attention_rpe.py:188-190)."""
    y, flag = y_flag
    return residual(x, y, p, training, zero_y=flag)


def transformer_block(block, x: Tensor, src_invalid: Tensor, n: int, S: int, self_knn, cross=None, p: float = 0.0,
                      training: bool = False) -> Tensor:
    """transformer_rpe.py:48-135,207-245. x [n*S,128]; self_knn = Targets kwargs (idx, invalid, emb | rel, freqs) among the sources;
    cross(layer) -> list[Targets] with UN-normalised tokens (norm_tgt is applied here)."""
    if _chains_ok(x):
        return _transformer_block_chains(block, x, src_invalid, n, S, self_knn, cross, p, training)
    ln = lambda m, t: layer_norm(t, m)
    inv = src_invalid.reshape(-1).to(torch.uint8)
    # the glue between the GEMMs / attention calls (zeroing of rows without a valid target, dropout, residual add, relu, the closing
    # row mask) as one pass per tensor: residual() / relu_drop() - the dropout sites keep the order of the reference's modules
    for layer in block.layers:
        if block.mode == "dec_cross_attn":
            s = ln(layer.norm_src, x)
            ts = Targets(s, n_tgt=S, **self_knn)
            x = _attn_residual(x, attention(layer.attn_src, s, [ts], [kv_table(layer.attn_src, None, ts)], n, S, raw=True), p, training)
            s2 = ln(layer.norm1, x)
            tg = list(cross(layer))
            x = _attn_residual(x, attention(layer.attn, s2, tg, [kv_table(layer.attn, layer.norm_tgt, t) for t in tg], n, S, raw=True), p, training)
        else:  # enc_self_attn: gathered targets share norm1 with the source
            s2 = ln(layer.norm1, x)
            ts = Targets(s2, n_tgt=S, **self_knn)
            x = _attn_residual(x, attention(layer.attn, s2, [ts], [kv_table(layer.attn, None, ts)], n, S, raw=True), p, training)
        h = relu_drop(linear(ln(layer.norm2, x), layer.linear1.weight, layer.linear1.bias), p, training)
        x = residual(x, linear(h, layer.linear2.weight, layer.linear2.bias), p, training, zero_out=inv)
    return x


# ------------------------------------------------------------------------------------------------ small modules
def mlp(m, x: Tensor, training: bool = False) -> Tensor:
    """modules/mlp.py:69-72 (Linear [+LN] [+ReLU] [+Dropout] per layer)."""
    p = m.dropout_p
    for lin, lnm, act in m.linear_layers():
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


POINTNET_FUSED = os.environ.get("TBX_POINTNET_FUSED", "1") != "0"


def _pointnet_fused_ok(enc, x: Tensor, training: bool) -> bool:
    if not (POINTNET_FUSED and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and 0 < x.shape[1] <= 16 and x.shape[0] > 0):
        return False
    for m in enc.mlp_layers:
        ll = m.linear_layers()
        if len(ll) != 1 or ll[0][1] is not None or not ll[0][2] or ll[0][0].weight.shape[0] != 64:
            return False
        if training and m.dropout_p > 0 and _DROP is None:  # torch's generator (unit tests of single modules): the plain ops
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
                _DROP["site"] += 1  # the id _drop would have given this layer's dropout
                rows = z.shape[0] * z.shape[1]
                assert rows % _DROP["n_batch"] == 0
                drop = (float(m.dropout_p), _DROP["seed"], _DROP["site"], rows // _DROP["n_batch"], _DROP["tb"], _DROP["t0"])
            x = PointNetTailFn.apply(z, inv8, *drop)
        return MaskedMaxPoolFn.apply(x, inv8)
    im = invalid.unsqueeze(-1)
    for m in enc.mlp_layers:
        h = mlp(m, x, training).masked_fill(im, float("-inf"))
        x = torch.cat([h, h.amax(dim=1, keepdim=True).expand(-1, h.shape[1], -1)], -1).masked_fill(im, 0.0)
    y = x.masked_fill(im, float("-inf")).amax(1)
    return y.masked_fill(invalid.all(-1, keepdim=True), 0.0)


def _knn(src_pose, src_inv, tgt_pose, tgt_inv, k, limit, rp, div=1):
    """-> kwargs of Targets: KNN indices / mask + the relative poses (the embedding is rebuilt inside the attention kernels)."""
    idx, inv, rel, _ = hip.knn_embed(src_pose, src_inv, tgt_pose, tgt_inv, k, limit, tgt_batch_div=div, want_rel_pose=True,
                                     want_emb=False)
    # with gradients on: the inverse lists the attention backward gathers dK / dV through
    lists = hip.knn_inverse(idx, inv, tgt_pose.shape[1], div) if torch.is_grad_enabled() else None
    return dict(idx=idx, invalid=inv, emb=None, rel=rel, freqs=(rp.pe_xy.freqs, rp.pe_yaw.freqs), inv=lists)


# ------------------------------------------------------------------------------------------------ encoders
def map_encoder(me, mp_valid, mp_attr, mp_pose, mp_type, training: bool) -> Dict[str, Tensor]:
    n, M, N = mp_valid.shape
    dev, d = mp_pose.device, me.hidden_dim
    rows = n * M * N
    attr = torch.empty(rows, 32, dtype=torch.float32, device=dev)
    pe = torch.empty(rows, 8, dtype=torch.float32, device=dev)
    row_inv = torch.empty(rows, dtype=torch.uint8, device=dev)
    tok_pose = torch.empty(n, M, 3, dtype=torch.float32, device=dev)
    tok_inv = torch.empty(n, M, dtype=torch.uint8, device=dev)
    hip.map_prep(mp_valid.to(torch.uint8).contiguous(), mp_attr.float().contiguous(), mp_pose.float().contiguous(), attr, pe, row_inv,
                 tok_pose, tok_inv)
    x = torch.cat([mlp(me.input_encoder.mlp, attr[:, :me.input_encoder.mlp.input_dim], training), pe[:, :7]], -1)
    feat = pointnet(me.pl_encoder, x.view(n * M, N, d), row_inv.view(n * M, N).bool(), training)
    knn = _knn(tok_pose, tok_inv, tok_pose, tok_inv, me.n_tgt_knn, me.dist_limit, me.pose_rpe)
    feat = transformer_block(me.tf_mp2mp, feat, tok_inv, n, M, knn, p=me.tf_mp2mp.dropout_p, training=training)
    return {"mp_token_invalid": tok_inv.bool(), "mp_token_invalid_u8": tok_inv, "mp_token_feature": feat.view(n, M, d),
            "mp_token_pose": tok_pose, "mp_token_type": mp_type}


def tl_pre_compute(te, tl_valid, tl_attr, tl_pose, mp: Dict[str, Tensor]) -> Dict[str, Tensor]:
    n, L = tl_valid.shape
    dev = tl_pose.device
    tl_inv = (~tl_valid).to(torch.uint8).contiguous()
    pose = tl_pose.float().contiguous()
    mpf = mp["mp_token_feature"].detach() if te.tl_lane_detach_mp_feature else mp["mp_token_feature"]
    t = {"tl_token_valid": tl_valid, "tl_token_invalid": ~tl_valid, "tl_token_invalid_u8": tl_inv, "tl_token_pose": pose,
         "tl_token_attr": mpf[torch.arange(n, device=dev).unsqueeze(1), tl_attr], "mp_feat_for_tl": mpf}
    t["tt"] = _knn(pose, tl_inv, pose, tl_inv, te.n_tgt_knn_tl2tl, te.dist_limit, te.pose_rpe)
    t["tm"] = _knn(pose, tl_inv, mp["mp_token_pose"], mp["mp_token_invalid_u8"], te.n_tgt_knn_tl2mp, te.dist_limit, te.pose_rpe)
    return t


def expand_tl_tokens(te, t: Dict[str, Tensor], mp: Dict[str, Tensor], T: int) -> Dict[str, Tensor]:
    """tl_pre_compute's per-scene dict for a time-batched call: batch entry b = scene b // T. Per-light tensors are repeated,
    the K-nearest sets are searched again on the repeated poses (identical sets; the map tables stay per scene: div = T)."""
    rep = lambda x: x.repeat_interleave(T, 0)
    e = {k: rep(t[k]) for k in ("tl_token_valid", "tl_token_invalid", "tl_token_invalid_u8", "tl_token_pose", "tl_token_attr")}
    e["tl_token_invalid_u8"], e["tl_token_pose"] = e["tl_token_invalid_u8"].contiguous(), e["tl_token_pose"].contiguous()
    e["mp_feat_for_tl"], e["time_batch"] = t["mp_feat_for_tl"], T
    e["tt"] = _knn(e["tl_token_pose"], e["tl_token_invalid_u8"], e["tl_token_pose"], e["tl_token_invalid_u8"], te.n_tgt_knn_tl2tl,
                   te.dist_limit, te.pose_rpe)
    e["tm"] = _knn(e["tl_token_pose"], e["tl_token_invalid_u8"], mp["mp_token_pose"], mp["mp_token_invalid_u8"], te.n_tgt_knn_tl2mp,
                   te.dist_limit, te.pose_rpe, div=T)
    e["_kv_cache"] = {}
    return e


def tl_encoder(te, hist_tl: Tensor, t: Dict[str, Tensor], training: bool) -> Tensor:
    """hist_tl [n,L,W] u8 masks (0xFF missing) -> [n*L, 128]. traffic_light.py:184-246. `t` = tl_pre_compute's dict, or
    expand_tl_tokens' (n = scenes x time batch, the map tokens shared by the T entries of a scene)."""
    n, L, W = hist_tl.shape
    dev, d = hist_tl.device, te.hidden_dim
    ld = 16 if 5 + W <= 16 else 32
    attr = torch.empty(n * L * W, ld, dtype=torch.float32, device=dev)
    row_inv = torch.empty(n * L * W, dtype=torch.uint8, device=dev)
    hip.tl_prep(hist_tl, t["tl_token_invalid_u8"], attr, row_inv)
    x = mlp(te.input_encoder.mlp, attr[:, :5 + W], training).view(n * L, W, d) + t["tl_token_attr"].reshape(n * L, 1, d)
    x = pointnet(te.temp_encoder, x, row_inv.view(n * L, W).bool(), training)
    M = t["mp_feat_for_tl"].shape[1]
    mp_tokens = t["mp_feat_for_tl"].reshape(-1, d)
    return transformer_block(te.tf_tl2tlmp, x, t["tl_token_invalid_u8"], n, L, t["tt"],
                             cross=lambda layer: [Targets(mp_tokens, n_tgt=M, cache=t.get("_kv_cache"), key="tl2mp",
                                                          batch_div=t.get("time_batch", 1), **t["tm"])],
                             p=te.tf_tl2tlmp.dropout_p, training=training)


def agent_encoder(ae, hist_valid, hist_pose, hist_motion, ag_attr6, mp, tl_inv_u8, tl_pose, tl_feat, training: bool, T: int = 1):
    """agent_encoder.py:114-178,321-387. hist_* [n,A,W(,3)] oldest first -> (feat [n*A,128], prep dict). T > 1: time-batched
    call, n = scenes x T, `mp` per scene (shared by a scene's T entries), the light tensors per entry."""
    n, A, W = hist_valid.shape
    dev, d = hist_pose.device, ae.hidden_dim
    prep = ae.alloc_prep(n, A, dev, with_heads=False)
    hip.agent_prep(hist_valid, hist_pose, hist_motion, ag_attr6, None, ae.pose_emb.pe_xy.freqs, ae.pose_emb.pe_yaw.freqs,
                   ae.pose_emb.out_dim, prep)
    tok_pose, tok_inv = prep["tok_pose"], prep["tok_invalid"]
    aa = _knn(tok_pose, tok_inv, tok_pose, tok_inv, ae.n_tgt_knn_ag2ag, ae.dist_limit, ae.pose_rpe)
    am = _knn(tok_pose, tok_inv, mp["mp_token_pose"], mp["mp_token_invalid_u8"], ae.n_tgt_knn_ag2mp, ae.dist_limit, ae.pose_rpe, div=T)
    at = _knn(tok_pose, tok_inv, tl_pose, tl_inv_u8, ae.n_tgt_knn_ag2tl, ae.dist_limit, ae.pose_rpe)
    x = torch.cat([mlp(ae.input_encoder.mlp, prep["attr"][:, :ae.input_encoder.mlp.input_dim], training), prep["pe"]], -1)
    x = pointnet(ae.temp_encoder, x.view(n * A, W, d), prep["row_invalid"].view(n * A, W).bool(), training)
    M, L = mp["mp_token_pose"].shape[1], tl_pose.shape[1]
    mp_tokens = mp["mp_token_feature"].reshape(-1, d)
    x = transformer_block(ae.tf_ag2agmptl, x, tok_inv, n, A, aa,
                          cross=lambda layer: [Targets(mp_tokens, n_tgt=M, cache=mp.get("_kv_cache"), key="ag2mp", batch_div=T, **am),
                                               Targets(tl_feat, n_tgt=L, **at)],
                          p=ae.tf_ag2agmptl.dropout_p, training=training)
    return x, prep


def add_navi_latent(m, x: Tensor, z: Tensor, z_invalid: Tensor, training: bool) -> Tensor:
    zi = z_invalid.unsqueeze(-1)
    zz = mlp(m.mlp_in, z, training).masked_fill(zi, 0.0)
    h = mlp(m.mlp, torch.cat([x, zz], -1), training).masked_fill(zi, 0.0)
    return h + x


def _agent_policy_engine(model, hv, hp, hm, ag_attr6, ag_type, z, z_valid, dest, navi_valid, tl_tokens, mp, tl_pre, training) -> Tensor:
    """The agents' half of a policy step on the inference engine's schedule (TrafficBots.agent_policy: window PointNet chain,
    3 K-nearest searches, 4 layer chains + 8 attention launches, 2 heads chains) - for the stepping pass of training: no
    autograd, light tokens given (tl_pre), keyed dropouts emitted as stages / in-kernel in the torch path's site order."""
    from . import engine

    n, A, _ = hv.shape
    tl_feat, ids = tl_pre[0], tl_pre[1]
    u8 = torch.uint8
    out = dict(action_mean=torch.empty(n * A, 2, dtype=torch.float32, device=hp.device))
    # (tl_pre[2]: this step's K/V tables, made for all steps of the piece in one launch - the same row-local stages)
    tl_kv = tl_pre[2] if len(tl_pre) > 2 and tl_pre[2] is not None else engine.kv_tables(tl_feat.contiguous(), model.ag_encoder.tl_kv_layers())
    ctx = None
    if training and _DROP is not None:
        ctx = dict(seed=_DROP["seed"], site=ids[0], call=ids[1], step=_DROP["t0"])
    # per-rollout constants in the form the engine takes them: converted once per pass, not in each of its 90 steps (the pass's
    # map dict lives as long as the pass: ~8 small launches per step less)
    sc = mp.setdefault("_step_consts", {})

    def const(name, t, fn):
        k = (name, t.data_ptr(), tuple(t.shape))
        if k not in sc:
            sc[k] = fn(t)
        return sc[k]

    type_idx = const("type", ag_type, lambda t: t.to(u8).argmax(-1).to(u8).contiguous())
    zz = const("z", z, lambda t: t.reshape(n * A, -1).float().contiguous())
    zi = const("zi", z_valid, lambda t: (~t).reshape(-1).to(u8).contiguous())
    dd = const("dest", dest, lambda t: t.contiguous())

    def dest_feature(_):  # mlp_mp(map feature of the destination): no dropout in it, fixed over the rollout (TrafficBots.rollout_constants)
        M = mp["mp_token_pose"].shape[1]
        rows_ = (torch.arange(n, device=dd.device).unsqueeze(1) * M + dd).reshape(-1).to(torch.int32).contiguous()
        dst = torch.empty(n * A, model.hidden_dim, dtype=torch.float32, device=dd.device)
        ch = hip.Chain(16, 4 * model.hidden_dim + 4)
        model.navi_encoder.emit_dest_feature(ch, mp["mp_token_feature"].reshape(-1, model.hidden_dim), rows_, dst)
        ch.run(n * A)
        return dst

    rc = dict(dest_feature=const("destf", dd, dest_feature)) if HEADS_TILE else None
    engine.DROP_CTX = ctx
    try:
        # (the stepping pass keeps the VALU ring kernel in both classes: at its 1024 source rows - one row per wavefront, one wavefront
        # per SIMD - the matrix-core form measured 20.0 us per launch against 17.1, profiles/r05b_train_replay_timeline_bf16.txt; the
        # two passes then differ by the bf16 operand rounding of the batched forward, ~1e-2 of an action, far below the dropout noise)
        nv8 = navi_valid.view(u8) if navi_valid.dtype == torch.bool else navi_valid.to(u8)
        model.agent_policy(hv, hp, hm, ag_attr6, type_idx, zz, zi, dd, nv8.contiguous(), tl_tokens, mp, tl_kv, out, rollout_consts=rc)
    finally:
        engine.DROP_CTX = None
    if ctx is not None:
        _DROP["site"], _DROP["call"] = ctx["site"], ctx["call"]
    return out["action_mean"].view(n, A, 2)


def policy_step(model, hist, ag_attr6, ag_type, ag_valid, ag_pose, z, z_valid, dest, navi_valid, tl_tokens, mp, training: bool,
                T: int = 1, tl_pre=None, want_logits: bool = True):
    """traffic_bots.py:188-221 -> (action mean [n,A,2], tl logits [n,L,5]). T > 1: time-batched call - every per-entry
    argument has n = scenes x T entries ([scene][step] order), `mp` stays per scene and `tl_tokens` is expand_tl_tokens'.
    tl_pre = (tl_feat [n*L,128], (site, call) dropout ids after the light encoder): the light tokens were encoded ahead.
    want_logits=False (with tl_pre, autograd off): only the action means are needed - the agents' half runs on the engine's
    chains (`_agent_policy_engine`) and None is returned for the logits."""
    hv, hp, hm, ht = hist
    n, A, W = hv.shape
    d = model.hidden_dim
    L = ht.shape[1]
    assert tl_tokens.get("time_batch", 1) == T
    if tl_pre is not None and T == 1 and not want_logits and _chains_ok(hp):
        return _agent_policy_engine(model, hv, hp, hm, ag_attr6, ag_type, z, z_valid, dest, navi_valid, tl_tokens, mp, tl_pre, training), None
    if tl_pre is None:
        tl_feat = tl_encoder(model.tl_encoder, ht, tl_tokens, training)
    else:
        tl_feat = tl_pre[0].reshape(-1, d)
        if _DROP is not None:  # the agents' dropout sites keep the ids they have when the light encoder runs in place
            _DROP["site"], _DROP["call"] = tl_pre[1]
    feat, _ = agent_encoder(model.ag_encoder, hv, hp, hm, ag_attr6, mp, tl_tokens["tl_token_invalid_u8"], tl_tokens["tl_token_pose"],
                            tl_feat, training, T)
    # NaviEncoder (navigation.py:65-79): detached map feature of the destination + pose embedding of its relative pose
    ne = model.navi_encoder
    bi = (torch.arange(n, device=feat.device) // T).unsqueeze(1)
    mpf = mp["mp_token_feature"].detach() if ne.dest_detach_mp_feature else mp["mp_token_feature"]
    gp = mp["mp_token_pose"][bi, dest]
    c, s = torch.cos(ag_pose[..., 2]), torch.sin(ag_pose[..., 2])
    dx, dy = gp[..., 0] - ag_pose[..., 0], gp[..., 1] - ag_pose[..., 1]
    rel = torch.stack([dx * c + dy * s, dx * (-s) + dy * c, gp[..., 2] - ag_pose[..., 2]], -1).reshape(-1, 3).contiguous()
    pe = hip.pose_embed(rel, ne.pose_emb.pe_xy.freqs, ne.pose_emb.pe_yaw.freqs, ne.pose_emb.out_dim)
    navi = mlp(ne.mlp_mp, mpf[bi, dest].reshape(n * A, d)) + mlp(ne.mlp_pe, pe)
    feat = add_navi_latent(model.add_navi, feat, navi, ~navi_valid.reshape(-1), training)
    feat = add_navi_latent(model.add_latent, feat, z.reshape(n * A, -1), ~z_valid.reshape(-1), training)
    mask_type = ~(ag_type & ag_valid.unsqueeze(-1)).reshape(n * A, 3)
    mean = 0
    for i, m in enumerate(model.action_head.mlp_mean):
        mean = mean + mlp(m, feat).masked_fill(mask_type[:, i:i + 1], 0.0)
    sp = model.tl_state_predictor
    xt = tl_feat.detach() if sp.detach_tl_feature else tl_feat
    logits = torch.clamp(mlp(sp.mlp, xt).masked_fill(tl_tokens["tl_token_invalid"].reshape(-1, 1), 0.0), -3, 3)
    return mean.view(n, A, 2), logits.view(n, L, -1)


def latent_posterior(le, b, mp, tl_tokens, training: bool) -> DiagGaussian:
    r = le.temporal_down_sample_rate
    v, m, p, s = b["gt/ag_valid"][:, :, ::r], b["gt/ag_motion"][:, :, ::r], b["gt/ag_pose"][:, :, ::r], b["gt/tl_state"][:, :, ::r]
    te, ae = le.tl_encoder_post, le.ag_encoder_post
    n, A, _ = v.shape
    tl_feat = tl_encoder(te, te.states_to_hist(s, te.temp_window_size), tl_tokens, training)
    hv, hp, hm = ae.pad_hist(v, p, m, ae.temp_window_size)
    feat, _ = agent_encoder(ae, hv, hp, hm, b["sc/ag_attr"].float().contiguous(), mp, tl_tokens["tl_token_invalid_u8"],
                            tl_tokens["tl_token_pose"], tl_feat, training)
    valid = b["gt/ag_valid"].any(-1)
    mean = mlp(le.latent_dist_post.mlp_mean, feat).view(n, A, -1).masked_fill(~valid.unsqueeze(-1), 0.0)
    return DiagGaussian(mean, le.latent_dist_post.log_std, valid=valid)


class NaviPairFirstLayer(torch.autograd.Function):
    """First Linear of NaviPredictor's pair MLP (navigation.py:245-262) without the [n, A, M, 384] concatenation:
        W [128, 384] = [W_a | W_m | W_e]:  h[n, a, m] = W_a f_a[n, a] + (W_m f_m[n, m] + b) + W_e e(rel[n, a, m])
    The per-agent and per-polyline terms are [n, A, 128] / [n, M, 128] GEMMs handed in; this function adds the per-pair term, with
    the 128-d pose embedding e rebuilt from the 12-byte relative pose scene by scene (tbx_pose_embed) in forward AND backward -
    neither the concatenation (1,536 B per pair) nor the embedding (512 B per pair) is kept for autograd: 12 B per pair are."""

    @staticmethod
    def forward(ctx, rel, w_e, pa, pm, fxy, fyw):
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
            h[i] += pa[i].unsqueeze(1) + pm[i].unsqueeze(0)
        ctx.save_for_backward(rel, w_e, fxy, fyw)
        return h

    @staticmethod
    def backward(ctx, dh):
        rel, w_e, fxy, fyw = ctx.saved_tensors
        n, A, M, _ = rel.shape
        dh = dh.contiguous()
        dw = torch.zeros(w_e.shape, dtype=torch.float32, device=dh.device)
        for i in range(n):
            emb = hip.pose_embed(rel[i].reshape(-1, 3), fxy, fyw, w_e.shape[1])
            dw += hip.linear_wgrad(dh[i].view(A * M, -1), emb, want_db=False, bf16=ctx.bf16)[0]  # dY^T X over 65 k rows: tbx_linear_wgrad
        return None, dw, dh.sum(2), dh.sum(1), None, None


def navi_predictor(npd, b, mp, training: bool) -> DestCategorical:
    """navigation.py:175-278 (dest)."""
    ag_valid, ag_pose, ag_motion = b["sc/ag_valid"], b["sc/ag_pose"].detach(), b["sc/ag_motion"].detach()
    n, A, W = ag_valid.shape
    assert W <= npd.temp_window_size
    dev, d = ag_pose.device, npd.hidden_dim
    from .models.agent_encoder import AgentEncoder

    hv, hp, hm = AgentEncoder.pad_hist(ag_valid, ag_pose, ag_motion, npd.temp_window_size)
    Wn = npd.temp_window_size
    f32, u8 = torch.float32, torch.uint8
    prep = dict(tok_pose=torch.empty(n, A, 3, dtype=f32, device=dev), tok_invalid=torch.empty(n, A, dtype=u8, device=dev),
                attr=torch.empty(n * A * Wn, 32, dtype=f32, device=dev), pe=torch.empty(n * A * Wn, npd.pose_emb.out_dim, dtype=f32, device=dev),
                row_invalid=torch.empty(n * A * Wn, dtype=u8, device=dev))
    hip.agent_prep(hv, hp, hm, b["sc/ag_attr"].float().contiguous(), None, npd.pose_emb.pe_xy.freqs, npd.pose_emb.pe_yaw.freqs,
                   npd.pose_emb.out_dim, prep)
    x = torch.cat([mlp(npd.input_encoder.mlp, prep["attr"][:, :npd.input_encoder.mlp.input_dim], training), prep["pe"]], -1)
    feat = pointnet(npd.temp_encoder, x.view(n * A, Wn, d), prep["row_invalid"].view(n * A, Wn).bool(), training)
    mpf = mp["mp_token_feature"].detach() if npd.detach_input else mp["mp_token_feature"]
    M = mpf.shape[1]
    tp, mpp = prep["tok_pose"], mp["mp_token_pose"]
    c, s = torch.cos(tp[..., 2])[:, :, None], torch.sin(tp[..., 2])[:, :, None]
    dx, dy = mpp[:, None, :, 0] - tp[:, :, None, 0], mpp[:, None, :, 1] - tp[:, :, None, 1]
    rel = torch.stack([dx * c + dy * s, dx * (-s) + dy * c, mpp[:, None, :, 2] - tp[:, :, None, 2]], -1).contiguous()  # [n, A, M, 3]
    # first Linear split into per-agent + per-polyline + per-pair terms: no [n, A, M, 384] tensor
    (lin1, ln1, act1), rest = npd.mlp.linear_layers()[0], npd.mlp.linear_layers()[1:]
    w1 = lin1.weight
    pa = linear(feat.view(n, A, d), w1[:, :d], None)
    pm = linear(mpf, w1[:, d:2 * d], lin1.bias)
    x = NaviPairFirstLayer.apply(rel, w1[:, 2 * d:], pa, pm, npd.pose_rpe.pe_xy.freqs, npd.pose_rpe.pe_yaw.freqs)
    if ln1 is not None:
        x = layer_norm(x, ln1)
    if act1:
        x = F.relu(x)
    x = _drop(x, npd.mlp.dropout_p, training)
    for lin, lnm, act in rest:
        x = linear(x, lin.weight, lin.bias)
        if lnm is not None:
            x = layer_norm(x, lnm)
        if act:
            x = F.relu(x)
        x = _drop(x, npd.mlp.dropout_p, training)
    logits = x.squeeze(-1)
    ty, ag_type = mp["mp_token_type"], b["ref/ag_type"]
    tok_valid = ag_valid.any(-1)
    mp_mask = mp["mp_token_invalid"] | ~(ty[:, :, :5].any(-1))
    bad = (mp_mask[:, None] | (ag_type[:, :, 0:1] & ty[:, :, 3][:, None]) | (ag_type[:, :, 1:2] & ty[:, :, :4].any(-1)[:, None])
           | (ag_type[:, :, 2:3] & ty[:, :, :3].any(-1)[:, None]))
    logits = logits.masked_fill(bad, float("-inf")).masked_fill((~tok_valid).unsqueeze(-1) | bad.all(-1, keepdim=True), 0)
    return DestCategorical(logits=logits, valid=tok_valid)


# ------------------------------------------------------------------------------------------------ rollout + loss
def _bits(one_hot: Tensor) -> Tensor:
    w = (1 << torch.arange(one_hot.shape[-1], device=one_hot.device, dtype=torch.int32))
    return (one_hot.to(torch.int32) * w).sum(-1).to(torch.uint8)


def training_rollout(wm, b, mp, tl_tokens, z, z_valid, tf_mask: Tensor, step_end: int, policy=None, record: Optional[dict] = None,
                     need_hist: bool = True) -> Dict[str, Tensor]:
    """Closed-loop training rollout (waymo_motion.py:206-311 with training=True: model inputs detached, the only
    cross-step gradient path is the dynamics chain). The per-step state machine is elementwise torch on [n,A] tensors
    (it needs autograd); rule feedback = outside-map + dest-reached as in tbx_sim_step.
    policy(step, hist, valid, pose, navi_valid) -> (mean [n,A,2], logits [n,L,5]); None = policy_step with autograd, one step
    at a time (the reference's order of evaluation). record: dict of lists that receives every step's policy inputs.
    need_hist=False: the policy does not read the windows (they are not built)."""
    model, dyn, rc = wm.model, wm.dynamics, wm.hp.differentiable_reward
    gt_valid, gt_pose, gt_motion, tl_gt = b["gt/ag_valid"], b["gt/ag_pose"], b["gt/ag_motion"], b["gt/tl_state"]
    ag_type, ag_attr6, dest = b["ref/ag_type"], b["sc/ag_attr"].float().contiguous(), b["gt/ag_navi"]
    n, A, Tg = gt_valid.shape
    L, Tt = tl_gt.shape[1], tl_gt.shape[2]
    W, dev = model.temp_window_size, gt_pose.device
    dt = dyn.dt
    if getattr(dyn, "_max_act", None) is None or dyn._max_act.device != dev:  # host -> device once (not inside a captured step)
        dyn._max_act = torch.tensor([[a, y] for a, y in zip(dyn.max_acc, dyn.max_yaw_rate)], device=dev)
    max_act = dyn._max_act
    lim = (ag_type.unsqueeze(-1) * max_act).sum(2)
    bi = torch.arange(n, device=dev).unsqueeze(1)
    d_type = b["map/type"][bi, dest]
    d_dir = b["map/dir"][bi, dest][..., :2]
    d_dir = d_dir / torch.norm(d_dir, dim=-1, keepdim=True)
    d_pos, d_inv = b["map/pos"][bi, dest][..., :2], ~b["map/valid"][bi, dest]
    d_thresh = 50.0 * (1 - d_type[:, :, 4].float() * 0.8)
    bnd = b["map/boundary"]
    valid, disabled = gt_valid[:, :, 0], torch.zeros_like(gt_valid[:, :, 0])
    pose, motion = gt_pose[:, :, 0], gt_motion[:, :, 0]
    tl_bits = _bits(tl_gt)
    navi_valid = gt_valid.any(-1)
    outside, reached = torch.zeros_like(valid), torch.zeros_like(valid)
    hv = torch.zeros(n, A, W, dtype=torch.uint8, device=dev)
    hp, hm = torch.zeros(n, A, W, 3, device=dev), torch.zeros(n, A, W, 3, device=dev)
    ht = torch.full((n, L, W), 0xFF, dtype=torch.uint8, device=dev)
    tl_cur = tl_bits[:, :, 0]
    out = {k: [] for k in ("pred_valid", "pred_pose", "pred_motion", "tl_nll", "tl_nll_invalid", "reward", "reward_valid", "tf")}
    if policy is None:
        def policy(step, hist, valid_, pose_, navi_valid_):
            with _DropScope(n, 1, step, restart=_POLICY_SITE0):
                return policy_step(model, hist, ag_attr6, ag_type, valid_, pose_, z, z_valid, dest, navi_valid_, tl_tokens, mp,
                                   model.training)
    for step in range(1, step_end + 1):
        hist = None
        if need_hist:
            hv = torch.cat([hv[:, :, 1:], valid.to(torch.uint8).unsqueeze(2)], 2)
            hp = torch.cat([hp[:, :, 1:], pose.detach().unsqueeze(2)], 2)
            hm = torch.cat([hm[:, :, 1:], motion.detach().unsqueeze(2)], 2)
            ht = torch.cat([ht[:, :, 1:], tl_cur.unsqueeze(2)], 2)
            hist = (hv.contiguous(), hp.contiguous(), hm.contiguous(), ht.contiguous())
        if record is not None:
            for k, v in (("hv", hist[0]), ("hp", hist[1]), ("hm", hist[2]), ("ht", hist[3]), ("valid", valid), ("pose", pose.detach()),
                         ("navi_valid", navi_valid)):
                record.setdefault(k, []).append(v)
        mean, logits = policy(step, hist, valid, pose.detach(), navi_valid)
        inv1 = ~valid.unsqueeze(-1)
        action = (torch.tanh(mean) * lim).masked_fill(inv1, 0)
        acc, yr = action[..., 0], action[..., 1]
        v_t, th_t = motion[..., 0] + 0.5 * dt * acc, pose[..., 2] + 0.5 * dt * yr
        pose = (pose + dt * torch.stack([v_t * torch.cos(th_t), v_t * torch.sin(th_t), yr], -1)).masked_fill(inv1, 0)
        motion = torch.stack([motion[..., 0] + dt * acc, acc, yr], -1).masked_fill(inv1, 0)
        pred_valid, pred_pose, pred_motion = valid, pose, motion
        if step < Tg:
            ov = tf_mask[:, :, step] & ~disabled
            valid = valid | ov
            pose = pose.masked_fill(ov.unsqueeze(-1), 0) + gt_pose[:, :, step].masked_fill(~ov.unsqueeze(-1), 0)
            motion = motion.masked_fill(ov.unsqueeze(-1), 0) + gt_motion[:, :, step].masked_fill(~ov.unsqueeze(-1), 0)
            ov_log = tf_mask[:, :, step]
        else:
            ov_log = torch.zeros_like(valid)
        with torch.no_grad():
            if need_hist:
                tl_cur = tl_bits[:, :, step] if step < Tt else (1 << logits.argmax(-1)).to(torch.uint8)
            x, y = pred_pose[..., 0], pred_pose[..., 1]
            out_now = ((x > bnd[:, 1:2]) | (x < bnd[:, 0:1]) | (y > bnd[:, 3:4]) | (y < bnd[:, 2:3])) & pred_valid
            outside = outside | out_now
            dd = torch.norm(pred_pose[:, :, None, :2] - d_pos, dim=-1).masked_fill(d_inv, float("inf"))
            pos_ok = (dd < d_thresh.unsqueeze(-1)).any(-1)
            hf = torch.stack([torch.cos(pred_pose[..., 2]), torch.sin(pred_pose[..., 2])], -1)
            rot_ok = ((hf.unsqueeze(2) * d_dir).sum(-1).masked_fill(d_inv, 0) > 0.8660254037844387).any(-1)
            reach_now = (~reached) & pred_valid & ((d_type[:, :, :4].any(-1) & pos_ok & rot_ok) | (d_type[:, :, 4] & pos_ok))
            reached = reached | reach_now
        if step < Tg:  # rewards.py:58-74
            g_valid, g_pose, g_motion = gt_valid[:, :, step], gt_pose[:, :, step], gt_motion[:, :, step]
            r_valid = pred_valid & g_valid
            e_pos = F.smooth_l1_loss(g_pose[..., :2], pred_pose[..., :2], reduction="none").sum(-1)
            e_rot = 0.5 * (1 - torch.cos(g_pose[..., 2] - pred_pose[..., 2]))
            e_spd = F.smooth_l1_loss(g_motion[..., 0], pred_motion[..., 0], reduction="none")
            rew = ((-rc.l_pos.weight * e_pos).masked_fill(~r_valid, 0) + (-rc.l_rot.weight * e_rot).masked_fill(~r_valid, 0)
                   + (-rc.l_spd.weight * e_spd).masked_fill(~r_valid, 0))
            dis = out_now & ~g_valid
        else:
            r_valid, rew, dis = pred_valid, torch.zeros_like(pred_pose[..., 0]), out_now
        if step < Tt:
            nll = -Categorical(logits=logits, validate_args=False).log_prob(tl_gt[:, :, step].max(-1)[1])
            nll_inv = tl_tokens["tl_token_invalid"]
        else:
            nll, nll_inv = torch.zeros_like(logits[..., 0]), torch.ones_like(tl_tokens["tl_token_invalid"])
        for k, v in (("pred_valid", pred_valid), ("pred_pose", pred_pose), ("pred_motion", pred_motion), ("tl_nll", nll),
                     ("tl_nll_invalid", nll_inv), ("reward", rew), ("reward_valid", r_valid), ("tf", ov_log)):
            out[k].append(v)
        disabled = disabled | dis
        valid = valid & ~dis
        navi_valid = navi_valid & ~reach_now
    return {k: torch.stack(v, 2) for k, v in out.items()}


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


def tl_nll_all_steps(logits: Tensor, tl_gt: Tensor, tl_invalid: Tensor):
    """The light-state NLL of every step at once (waymo_motion.py:277-283): logits [n,T,L,5] of steps 1..T, tl_gt [n,L,Tt,5] one-hot,
    -> (nll [n,L,T], invalid [n,L,T]); steps without ground truth count as invalid."""
    n, T, L, _ = logits.shape
    S = min(T, tl_gt.shape[2] - 1)
    nll = torch.zeros(n, T, L, device=logits.device)
    inv = torch.ones(n, T, L, dtype=torch.bool, device=logits.device)
    if S > 0:
        idx = tl_gt[:, :, 1:S + 1].max(-1)[1].permute(0, 2, 1)  # [n,S,L]
        nll[:, :S] = -torch.log_softmax(logits[:, :S], -1).gather(-1, idx.unsqueeze(-1)).squeeze(-1)
        inv[:, :S] = tl_invalid.unsqueeze(1)
    return nll.permute(0, 2, 1), inv.permute(0, 2, 1)


# Pieces of the ahead-of-time light encoder (1: all steps at once on the one stream). Measured (bench.py --mode train): 1 -> 68.8,
# 2 -> 68.7, 3 -> 67.1, 5 -> 65.5 scenes/s - the light encoder's full-chip launches on the side stream do not fill the CUs the
# stepping pass leaves idle, they queue in front of its small launches (no preemption), so the overlap loses. Kept as an opt-in.
TL_CHUNKS = int(os.environ.get("TBX_TL_CHUNKS", "1"))
_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def training_rollout_batched(wm, b, mp, tl_tokens, z, z_valid, tf_mask: Tensor, step_end: int) -> Dict[str, Tensor]:
    """The same rollout, time-batched (module docstring): a step-by-step pass without autograd that records every step's
    policy inputs, then the T policy evaluations of every scene as one differentiated batch of n x T entries in [scene][step]
    order, then the per-step dynamics / reward chain on the batched means (the only cross-step gradient path)."""
    model = wm.model
    T = step_end
    n = b["gt/ag_valid"].shape[0]
    flat = lambda x: x.reshape(n * T, *x.shape[2:]).contiguous()
    rep = lambda x: x.repeat_interleave(T, 0)
    ag_attr6, ag_type, dest = b["sc/ag_attr"].float().contiguous(), b["ref/ag_type"], b["gt/ag_navi"]
    tl_T = expand_tl_tokens(model.tl_encoder, tl_tokens, mp, T)
    # The lights are teacher-forced while ground truth lasts (tl_cur = gt state for step < Tt), so with step_end <= Tt every
    # light window is known before the rollout: the light encoder of all T steps runs ONCE, differentiated, ahead of pass 1,
    # which then only steps the agents' half of the policy on its (detached) tokens.
    tl_gt = b["gt/tl_state"]
    L, Tt, W = tl_gt.shape[1], tl_gt.shape[2], model.temp_window_size
    tl_pre, tl_steps, ht_all = None, None, None
    if step_end <= Tt:
        pad = torch.full((n, L, W), 0xFF, dtype=torch.uint8, device=tl_gt.device)
        ht_all = torch.cat([pad, _bits(tl_gt)[:, :, :T]], 2).unfold(2, W, 1)[:, :, 1:T + 1]  # [n, L, T, W]: window of step s = states s-W .. s-1
        ht_all = ht_all.permute(0, 2, 1, 3).reshape(n * T, L, W).contiguous()
    tl_chunks = None  # [(first step index, [n, Tc, L, d] detached features, event or None)]
    if ht_all is not None and getattr(wm, "tl_encoder_ahead", True):
        # ... in TL_CHUNKS pieces along time: the first on this stream, the others on a side stream WHILE the stepping pass below runs
        # (its launches are latency-bound on a quarter of the CUs; the light encoder's are the big-batch kind). The stepping pass waits
        # for a piece's event before its first step. Same keyed masks (they are keyed by the absolute step), same per-row arithmetic.
        n_chunk = TL_CHUNKS if (ht_all.is_cuda and T >= 3 * TL_CHUNKS) else 1
        bounds = [round(i * T / n_chunk) for i in range(n_chunk + 1)]
        main, side = (torch.cuda.current_stream(), _side_stream(ht_all.device)) if n_chunk > 1 else (None, None)
        ht4, shared_kv, feats, tl_chunks, ids = ht_all.view(n, T, L, W), {}, [], [], None
        for c in range(n_chunk):
            c0, Tc = bounds[c], bounds[c + 1] - bounds[c]
            if c == 1:
                side.wait_stream(main)  # fork: the map K/V tables of the lights' cross attention were made (and cached) by piece 0
            with torch.cuda.stream(side) if c > 0 else contextlib.nullcontext():
                tl_Tc = tl_T
                if n_chunk > 1:
                    tl_Tc = expand_tl_tokens(model.tl_encoder, tl_tokens, mp, Tc)
                    tl_Tc["_kv_cache"] = shared_kv
                with _DropScope(n * Tc, Tc, 1 + c0, restart=_POLICY_SITE0):
                    f = tl_encoder(model.tl_encoder, ht4[:, c0:c0 + Tc].reshape(n * Tc, L, W).contiguous(), tl_Tc, model.training)
                    ids = (_DROP["site"], _DROP["call"]) if _DROP is not None else None
                ev = None
                if c > 0:
                    ev = torch.cuda.Event()
                    ev.record()
            feats.append(f.view(n, Tc, L, -1))
            tl_chunks.append((c0, feats[-1].detach(), ev))
    fused = getattr(wm, "fused_train_chain", True) and ht_all is not None  # (lights from their own logits: the torch state machine)
    chain = TrainChain(wm, b, tf_mask, T) if fused else None
    rec: Dict[str, List[Tensor]] = {}
    with torch.no_grad():  # pass 1: own K/V caches (its graph-less tables must not reach the differentiated pass)
        mp1, tl1 = dict(mp, _kv_cache={}), dict(tl_tokens, _kv_cache={})

        kv_pieces: Dict[int, Tensor] = {}

        def tl_of(step):  # the light tokens of `step` from the piece that holds it (joined on first use) + their K/V tables
            for i in range(len(tl_chunks) - 1, -1, -1):
                c0, f, ev = tl_chunks[i]
                if step - 1 >= c0:
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                        tl_chunks[i] = (c0, f, None)
                    if i not in kv_pieces and NOGRAD_CHAINS and f.is_cuda and not torch.is_grad_enabled():
                        # the K/V rows the agents' layers read, for ALL steps of the piece in one launch (time-major: a step's
                        # tables are then a contiguous [n * L, 1024] block) instead of one launch per step (90 x 37 us)
                        from . import engine

                        ft = f.permute(1, 0, 2, 3).contiguous()  # [Tc, n, L, d]
                        kv_pieces[i] = engine.kv_tables(ft.view(-1, ft.shape[-1]), model.ag_encoder.tl_kv_layers()).view(ft.shape[0], n * L, -1)
                    kv = kv_pieces[i][step - 1 - c0] if i in kv_pieces else None
                    # (with the tables at hand the engine path does not read the features: no per-step gather of them)
                    return (f[:, step - 1 - c0].reshape(n * L, -1) if kv is None else f[:, step - 1 - c0]), kv

        def policy1(step, hist, valid_, pose_, navi_valid_):
            pre = None
            if tl_chunks is not None:
                tf_, kv_ = tl_of(step)
                pre = (tf_, ids, kv_)
            with _DropScope(n, 1, step, restart=_POLICY_SITE0):
                return policy_step(model, hist, ag_attr6, ag_type, valid_, pose_, z.detach(), z_valid, dest, navi_valid_, tl1, mp1,
                                   model.training, tl_pre=pre, want_logits=not fused)

        if fused:  # the state machine is tbx_train_chain: per step the policy, then ONE launch
            ht_steps = ht_all.view(n, T, L, W)
            for step in range(1, T + 1):
                hv, hp, hm, valid_, pose_, navi_valid_ = chain.before(step)
                # (the light windows are only read when the lights are encoded in the step: not with their tokens made ahead)
                mean1, _ = policy1(step, (hv, hp, hm, ht_steps[:, step - 1] if tl_chunks is not None else ht_steps[:, step - 1].contiguous()),
                                   valid_, pose_, navi_valid_)
                chain.step(step, mean1)
            inputs = chain.windows()
        else:
            training_rollout(wm, b, mp1, tl1, z.detach(), z_valid, tf_mask, step_end, policy=policy1, record=rec)
            st = {k: torch.stack(v, 1) for k, v in rec.items()}  # [n, T, ...]
            inputs = (flat(st["hv"]), flat(st["hp"]), flat(st["hm"]), flat(st["valid"]), flat(st["pose"]), flat(st["navi_valid"]))
    if tl_chunks is not None:
        if len(tl_chunks) > 1:
            torch.cuda.current_stream().wait_stream(_side_stream(ht_all.device))  # join
        tl_feat_all = feats[0] if len(feats) == 1 else torch.cat(feats, 1)
        tl_pre = (tl_feat_all.reshape(n * T * L, -1), ids)
    hv, hp, hm, valid_all, pose_all, navi_all = inputs
    ht_in = ht_all if ht_all is not None else flat(st["ht"])
    with _DropScope(n * T, T, 1, restart=_POLICY_SITE0):
        mean, logits = policy_step(model, (hv, hp, hm, ht_in), rep(ag_attr6).contiguous(), rep(ag_type), valid_all, pose_all, rep(z),
                                   rep(z_valid), rep(dest), navi_all, tl_T, mp, model.training, T=T, tl_pre=tl_pre)
    mean, logits = mean.view(n, T, *mean.shape[1:]), logits.view(n, T, *logits.shape[1:])
    if fused:
        ro = chain.outputs(TrainChainFn.apply(mean, chain))
        ro["tl_nll"], ro["tl_nll_invalid"] = tl_nll_all_steps(logits, tl_gt, tl_tokens["tl_token_invalid"])
        return ro
    return training_rollout(wm, b, mp, tl_tokens, z, z_valid, tf_mask, step_end, need_hist=False,
                            policy=lambda step, hist, v, p, nv: (mean[:, step - 1], logits[:, step - 1]))


def training_loss(cfg, ro, navi_pred: DestCategorical, navi_gt, post: DiagGaussian, prior: DiagGaussian) -> Dict[str, Tensor]:
    """metrics/training.py:74-189 + metrics/loss.py:39-77 (default weights / switches)."""
    # A term whose counter is zero (a batch without a valid light / agent) is left out by the reference (training.py:166-186:
    # `if counter > 0`): sum = 0 over a count clamped to 1 gives that 0 without a host branch (the step stays capturable).
    cnt = lambda m: m.sum().clamp(min=1)
    lv = ro["pred_valid"].clone()
    lv[:, :, : cfg.step_training_start] &= False
    if not cfg.loss_for_teacher_forcing:
        lv &= ~ro["tf"]
    any_valid = lv.any(-1)
    P, Q = post.distribution, prior.distribution
    dP = Independent(Normal(P.base_dist.loc.detach(), P.base_dist.scale.detach(), validate_args=False), 1, validate_args=False)
    dQ = Independent(Normal(Q.base_dist.loc.detach(), Q.base_dist.scale.detach(), validate_args=False), 1, validate_args=False)
    e0 = torch.clamp(kl_divergence(dP, Q), min=cfg.kl_free_nats)
    e1 = torch.clamp(kl_divergence(P, dQ), min=cfg.kl_free_nats)
    kv = (post.valid if cfg.kl_for_unseen_agent else prior.valid) & any_valid
    vae_kl = cfg.w_vae_kl * (e0 + cfg.kl_balance_scale * e1).masked_fill(~kv, 0).sum() / cnt(kv)
    rv = lv & ro["reward_valid"]
    reward = cfg.w_diffbar_reward * ro["reward"].masked_fill(~rv, 0).sum() / cnt(rv)
    nv = navi_pred.valid & any_valid
    navi = cfg.w_navi * (-navi_pred.log_prob(navi_gt)).masked_fill(~nv, 0).sum() / cnt(nv)
    tv = ~ro["tl_nll_invalid"]
    tl = cfg.w_tl_state * ro["tl_nll"].masked_fill(~tv, 0).sum() / cnt(tv)
    return {"loss": vae_kl - reward + navi + tl, "vae_kl": vae_kl, "diffbar_reward": reward, "navi_loss": navi, "tl_state_loss": tl}


def training_step(wm, raw_batch: Dict[str, Tensor], noise: Optional[Tensor] = None, use_prior: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """waymo_motion.py:313-385. Dropout: residual / FFN / MLP dropouts through torch, the attention-probability dropout inside
    the HIP attention kernels (seed on the device, one call id per attention call of the step). `noise` [n,A,latent] / `use_prior` (0-d bool tensor) are the two host-drawn random
    inputs of a step as device tensors: a captured step (pl_modules/data_parallel.GraphedTrainStep) refills them before
    every replay; left None they are drawn here from the CPU generator like the reference's CPU path does."""
    global _FOLD_CACHE, _DROP, _PREC, _KV16
    _FOLD_CACHE, _KV16 = {}, {}
    _PREC = getattr(wm, "train_precision", None)
    if _PREC not in (None, "bf16", "fp32"):
        raise ValueError(f"train_precision {_PREC!r}: 'bf16' (autocast-class contractions) or 'fp32'")
    if wm.model.training:
        seed = getattr(wm, "attn_dropout_seed", None)  # a captured step owns a static seed tensor and refills it per replay
        if seed is None:
            seed = torch.empty(1, dtype=torch.int64, device=next(wm.model.parameters()).device).random_()
        _DROP = {"seed": seed, "call": 0, "site": 0, "n_batch": 0, "tb": 1, "t0": 0}
    hip.PACK_SCOPE = {}  # chain kernels of the stepping pass: weight images packed once per step (inside a captured step too)
    try:
        return _training_step(wm, raw_batch, noise, use_prior)
    finally:
        _FOLD_CACHE, _DROP, hip.PACK_SCOPE, _PREC, _KV16 = None, None, None, None, None


def _training_step(wm, raw_batch, noise, use_prior) -> Dict[str, Tensor]:
    model, hp = wm.model, wm.hp
    tr = model.training
    if "sc/mp_valid" in raw_batch:  # already re-keyed (a captured step pre-processes eagerly: its index tensors come from the host)
        b = raw_batch
    else:
        with torch.no_grad():
            b = wm.pre_processing(raw_batch)
    if _DROP is not None:
        _DROP["n_batch"] = b["sc/mp_valid"].shape[0]
    mp = map_encoder(model.mp_encoder, b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"], tr)
    tl_tokens = tl_pre_compute(model.tl_encoder, b["gt/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], mp)
    mp["_kv_cache"], tl_tokens["_kv_cache"] = {}, {}  # map K/V tables: once per training step, shared by all 90 steps
    post = latent_posterior(model.latent_encoder, b, mp, tl_tokens, tr)
    pr = model.latent_encoder.latent_dist_prior
    valid_hist = b["sc/ag_valid"].any(-1)
    prior = DiagGaussian(pr.mean.expand(*valid_hist.shape, -1), pr.log_std, valid=valid_hist)
    # rsample with the noise drawn from the CPU generator, as the reference's CPU path does (same stream under the same
    # seed); one [n, A, 16] host-to-device copy per training step
    if use_prior is None:
        lat = prior if torch.rand(1) < hp.p_training_rollout_prior else post
        l_mean, l_std, l_valid = lat.mean, lat.stddev, lat.valid
    else:  # the same choice as a device-side select (no host branch inside a captured step)
        l_mean = torch.where(use_prior, prior.mean, post.mean)
        l_std = torch.where(use_prior, prior.stddev.expand_as(post.mean), post.stddev.expand_as(post.mean))
        l_valid = torch.where(use_prior, prior.valid, post.valid)
    if noise is None:
        noise = torch.randn(l_mean.shape).to(l_mean.device)
    z = l_mean + l_std * noise
    navi_pred = navi_predictor(model.navi_predictor, b, mp, tr)
    tf = wm.teacher_forcing_training
    tf.init(ag_valid=b["gt/ag_valid"], ag_pose=b["gt/ag_pose"], ag_motion=b["gt/ag_motion"], tl_state=b["gt/tl_state"],
            current_epoch=wm.current_epoch)
    rollout = training_rollout_batched if getattr(wm, "time_batched_training", True) else training_rollout
    ro = rollout(wm, b, mp, tl_tokens, z, l_valid, tf.ag_teacher_forcing, hp.time_step_end)
    return training_loss(hp.training_metrics, ro, navi_pred, b["gt/ag_navi"], post, prior)
