"""Data-parallel training step: one process per GPU, scenes sharded across ranks, ONE exchange step per optimizer
step = gradient all-reduce over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

Replaces Lightning's `strategy="ddp"` (run.py:50-52) for this path. Differences that matter on MI355X:
  * the 2,745,030 parameters that never receive a gradient (std-normal prior copies, unused norm_tgt, action-head log_std:
    SURVEY.md finding 3) are excluded statically - no per-iteration unused-parameter graph walk;
  * all live gradients (~31.6 MB fp32) travel as ONE flat all-reduce after backward. With 8 fully connected xGMI
    peers a single large message amortises launch latency best; it is ~1 % of a training step, so it is not overlapped
    with backward (nothing to hide).
"""
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def live_parameters(module: torch.nn.Module) -> List[torch.nn.Parameter]:
    """Parameters that hold a gradient after a backward pass (call after the first backward)."""
    return [p for p in module.parameters() if p.requires_grad and p.grad is not None]


def allreduce_gradients(params: Iterable[torch.nn.Parameter], world_size: Optional[int] = None, group=None) -> int:
    """Average the gradients of `params` across ranks in place with one flat all-reduce. Returns the bytes exchanged."""
    params = [p for p in params if p.grad is not None]
    if not params or not dist.is_available() or not dist.is_initialized():
        return 0
    world_size = world_size or dist.get_world_size(group)
    if world_size == 1:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat.numel() * flat.element_size()


def train_step(wm, optimizer: torch.optim.Optimizer, batch: Dict[str, Tensor], clip_grad_norm: float = 5.0,
               live: Optional[List[torch.nn.Parameter]] = None) -> Dict[str, Tensor]:
    """fwd + bwd + gradient all-reduce + clip (trainer/default.yaml:13 gradient_clip_val 5.0) + optimizer step."""
    optimizer.zero_grad(set_to_none=True)
    loss = wm.training_step(batch, 0)
    loss.backward()
    params = live if live is not None else live_parameters(wm.model)
    allreduce_gradients(params)
    if clip_grad_norm and clip_grad_norm > 0:
        torch.nn.utils.clip_grad_norm_(params, clip_grad_norm)
    optimizer.step()
    return wm.last_metrics
