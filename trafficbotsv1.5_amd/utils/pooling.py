"""`seq_pooling` with the reference's signature (utils/pooling.py:7-38). "max_valid" - the mode every default module uses
(PolylineEncoder, sim_agent.yaml) - is the HIP masked max pool (tbx_masked_maxpool_fwd / _bwd, differentiable); "first" / "last" are
slices. "last_valid" / "mean_valid" are not used by any default entry point."""
from typing import Optional

import torch
from torch import Tensor

from .. import hip


class _MaskedMaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inv8):
        ctx.save_for_backward(x, inv8)
        return hip.masked_maxpool_fwd(x, inv8)

    @staticmethod
    def backward(ctx, dy):
        x, inv8 = ctx.saved_tensors
        return hip.masked_maxpool_bwd(dy.contiguous(), x, inv8), None


def seq_pooling(x: Tensor, invalid: Tensor, mode: str, valid: Optional[Tensor] = None) -> Tensor:
    """x [n_sc, n_ag, n_step, hidden], invalid [n_sc, n_ag, n_step] -> [n_sc, n_ag, hidden]; rows without a valid step -> 0."""
    n, A, W, d = x.shape
    if mode == "max_valid":
        inv8 = invalid.to(torch.uint8).contiguous().view(n * A * W)
        xf = x.float().reshape(n * A, W, d)
        # the kernel pools 128-column rows (hidden_dim of every default module); other widths go through it in 128-column blocks
        blocks = []
        for c0 in range(0, d, 128):
            xb = xf[..., c0:c0 + 128]
            if xb.shape[-1] < 128:
                xb = torch.nn.functional.pad(xb, (0, 128 - xb.shape[-1]))
            blocks.append(_MaskedMaxPool.apply(xb.contiguous(), inv8))
        y = blocks[0] if len(blocks) == 1 else torch.cat(blocks, -1)
        return y[:, :d].reshape(n, A, d)
    if mode in ("first", "last"):
        y = x[:, :, 0 if mode == "first" else -1]
        return y.masked_fill(invalid.all(-1, keepdim=True), 0)
    raise NotImplementedError(f"seq_pooling mode {mode} is not on the default hot path")
