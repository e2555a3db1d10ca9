"""Distribution wrappers with the reference's interface (modules/distributions.py): thin torch.distributions
holders, used once per batch (latent / destination sampling), never inside the per-step loop."""
from typing import Optional, Union

from torch import Tensor
from torch.distributions import Categorical, Independent, Normal


class MyDist:
    distribution = None
    valid: Optional[Tensor] = None


class DiagGaussian(MyDist):
    def __init__(self, mean: Tensor, log_std: Tensor, valid: Optional[Tensor] = None) -> None:
        self.mean, self.valid = mean, valid
        self.distribution = Independent(Normal(mean, log_std.exp(), validate_args=False), 1, validate_args=False)
        self.stddev = self.distribution.stddev

    def log_prob(self, sample: Tensor) -> Tensor:
        return self.distribution.log_prob(sample)

    def sample(self, deterministic: Union[bool, Tensor]) -> Tensor:
        if isinstance(deterministic, Tensor):
            d = deterministic.unsqueeze(-1)
            return self.distribution.mean.masked_fill(~d, 0) + self.distribution.rsample().masked_fill(d, 0)
        return self.distribution.mean if deterministic else self.distribution.rsample()

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        self.mean = self.mean.repeat_interleave(repeats, dim)
        self.stddev = self.stddev.repeat_interleave(repeats, dim)
        self.distribution = Independent(Normal(self.mean, self.stddev, validate_args=False), 1, validate_args=False)
        if self.valid is not None:
            self.valid = self.valid.repeat_interleave(repeats, dim)


class DestCategorical(MyDist):
    def __init__(self, probs: Optional[Tensor] = None, logits: Optional[Tensor] = None, valid: Optional[Tensor] = None):
        self.distribution = (Categorical(logits=logits, validate_args=False) if probs is None
                             else Categorical(probs=probs, validate_args=False))
        self.probs, self.valid = self.distribution.probs, valid

    def log_prob(self, sample: Tensor) -> Tensor:
        return self.distribution.log_prob(sample)

    def sample(self, deterministic: Union[bool, Tensor]) -> Tensor:
        if isinstance(deterministic, Tensor):
            return self.probs.argmax(-1).masked_fill(~deterministic, 0) + self.distribution.sample().masked_fill(deterministic, 0)
        return self.probs.argmax(-1) if deterministic else self.distribution.sample()

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        self.probs = self.probs.repeat_interleave(repeats, dim)
        self.distribution = Categorical(probs=self.probs, validate_args=False)
        if self.valid is not None:
            self.valid = self.valid.repeat_interleave(repeats, dim)
