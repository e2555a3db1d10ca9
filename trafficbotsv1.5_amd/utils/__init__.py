"""Host-side helpers that mirror `src/utils` of the reference where the hot path needs them."""
import zlib

import torch
from torch import nn


@torch.no_grad()
def det_fill(module: nn.Module, seed: int = 0) -> None:
    """Deterministic, name-keyed weight fill used by the parity fixtures.

    Every floating-point *parameter* is overwritten from a generator seeded by (seed, crc32(name)), so a
    reference model and this build's model (same state-dict keys and shapes) end up with identical weights
    without shipping a 42 MB checkpoint. Buffers (`freqs`, `hist_ohe`, `pl_node_ohe`) and frozen
    parameters (std-normal prior `mean` / `log_std`) keep their constructor values.
    """
    for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
        if not p.requires_grad or not p.is_floating_point():
            continue
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**31))
        leaf = name.rsplit(".", 1)[-1]
        if "log_std" in name:
            continue
        if p.dim() >= 2:
            fan_in = p.shape[-1]
            v = torch.randn(p.shape, generator=g) * (1.0 / fan_in**0.5)
        elif "norm" in name and leaf == "weight":
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif leaf == "weight":  # LayerNorm inside nn.Sequential MLPs (fc_layers.N.weight, 1-D)
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            v = 0.1 * torch.randn(p.shape, generator=g)
        p.copy_(v.to(p.device, p.dtype))
