"""`AngularError` / `BalancedKL` with the reference's interfaces (models/metrics/loss.py:9-77): the scalar glue of the training
loss. The reward's heading term inside a rollout is tbx_sim_step's / tbx_train_chain's; these classes serve callers that hold
tensors (a few hundred elements, once per batch - SURVEY 8a rows 18-19: negligible)."""
from typing import Optional

import torch
from torch import Tensor
from torch.distributions import Independent, Normal, kl_divergence


class AngularError:
    def __init__(self, criterion: str, angular_type: Optional[str]) -> None:
        if angular_type != "cosine":
            raise NotImplementedError("AngularError: the default configuration uses angular_type = cosine (sim_agent.yaml)")
        self.angular_type = angular_type

    def compute(self, preds: Tensor, target: Tensor) -> Tensor:
        return 0.5 * (1 - torch.cos(preds - target))


class BalancedKL:
    """Dreamer-v2 KL balancing (loss.py:39-77) for the diagonal-Gaussian latents of the default configuration."""

    def __init__(self, kl_balance_scale: float, kl_free_nats: float) -> None:
        self.alpha, self.free_nats = kl_balance_scale, kl_free_nats

    def compute(self, posterior: Independent, prior: Independent) -> Tensor:
        if not isinstance(posterior.base_dist, Normal):
            raise NotImplementedError("BalancedKL: diagonal-Gaussian latents (latent_encoder diag_gaus / std_gaus)")
        det = lambda d: Independent(Normal(d.base_dist.loc.detach(), d.base_dist.scale.detach(), validate_args=False), 1, validate_args=False)
        if self.alpha > 0:
            e0, e1 = kl_divergence(det(posterior), prior), kl_divergence(posterior, det(prior))
            if self.free_nats > 0:
                e0, e1 = torch.clamp(e0, min=self.free_nats), torch.clamp(e1, min=self.free_nats)
            return e0 + self.alpha * e1
        e = kl_divergence(posterior, prior)
        return torch.clamp(e, min=self.free_nats) if self.free_nats > 0 else e
