"""`PositionalEmbedding` / `PositionalEmbeddingRad` under the reference's module path (utils/positional_emb.py:6-54): the buffer
holders of utils/pose_emb.py plus their stand-alone `forward`, evaluated by tbx_pose_embed (the pe_xy_yaw layout contains both
embeddings as column blocks: [cos(x f) | sin(x f) | cos(y f) | sin(y f) | cos(k yaw) | sin(k yaw)], pose_emb.py:50-55)."""
import torch
from torch import Tensor

from .. import hip
from .pose_emb import PositionalEmbedding as _PE, PositionalEmbeddingRad as _PERad


def _rad_freqs(n: int, dev) -> Tensor:
    return (torch.arange(0, n // 2, device=dev) + 1.0).repeat_interleave(2, 0)


class PositionalEmbedding(_PE):
    def forward(self, x: Tensor) -> Tensor:
        """x [...] -> [..., dim] = [cos(x f_0..), sin(x f_0..)]; dim in {16, 32} (pe_dim 64 / 128 of the pose embedding)."""
        if self.dim not in (16, 32):
            raise NotImplementedError("stand-alone PositionalEmbedding.forward: dim 16 or 32 (pe_dim 64 / 128)")
        P = 4 * self.dim
        pose3 = torch.stack([x.float().reshape(-1), torch.zeros(x.numel(), device=x.device), torch.zeros(x.numel(), device=x.device)], -1).contiguous()
        out = hip.pose_embed(pose3, self.freqs, _rad_freqs(P // 2, x.device), P)
        return out[:, : self.dim].reshape(*x.shape, self.dim)


class PositionalEmbeddingRad(_PERad):
    def forward(self, x: Tensor) -> Tensor:
        """x [...] in rad -> [..., dim] = [cos(k x), sin(k x)], k = 1..dim/2; dim in {32, 64}."""
        if self.dim not in (32, 64):
            raise NotImplementedError("stand-alone PositionalEmbeddingRad.forward: dim 32 or 64 (pe_dim 64 / 128)")
        P = 2 * self.dim
        z = torch.zeros(x.numel(), device=x.device)
        pose3 = torch.stack([z, z, x.float().reshape(-1)], -1).contiguous()
        fxy = torch.ones(P // 4, device=x.device)  # (x = y = 0: the xy blocks are not read back)
        out = hip.pose_embed(pose3, fxy, self.freqs, P)
        return out[:, P // 2:].reshape(*x.shape, self.dim)
