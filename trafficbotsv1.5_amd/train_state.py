"""Shared state of the training schedule (train_ops.py / train_graph.py read and write it as `train_state.X`): the arithmetic class of the
step's contractions, the keyed-dropout scope, the per-step caches (folded attention weights, bfloat16 copies of K/V tables) and the
switches of the fused paths. One training step at a time per process (one process per GPU)."""
import contextlib
import os
from typing import Optional

import torch


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


WGRAD_MIN_ROWS = int(os.environ.get("TBX_WGRAD_MIN_ROWS", "16384"))  # from here on dW = dY^T X is a reduction over so many rows that the library GEMM has 1-2 output tiles


# K/V tables of the current training step that also exist as bfloat16 (made by the launch that made the fp32 rows: tbx_tall_linear_dual):
# fp32 table's data_ptr -> (the fp32 table - kept alive, so the address stays its own -, the bfloat16 copy). None outside a step.
_KV16: Optional[dict] = None


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


LN_BWD = os.environ.get("TBX_LN_BWD", "1") != "0"


LN_FWD = os.environ.get("TBX_LN_FWD", "1") != "0"


GLUE_FUSED = os.environ.get("TBX_GLUE_FUSED", "1") != "0"


# The stepping pass of the time-batched rollout needs no autograd: with gradients off, whole layers run as the inference
# engine's chain kernels (engine.run_block: ~6 launches per transformer layer instead of ~40 torch ops), with the keyed dropouts
# of training as DROPOUT stages / inside the attention kernels - same site / call ids, hence same masks, as the torch ops here.
NOGRAD_CHAINS = True


POINTNET_FUSED = os.environ.get("TBX_POINTNET_FUSED", "1") != "0"
