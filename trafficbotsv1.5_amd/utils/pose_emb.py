"""`PoseEmb` / `PositionalEmbedding[Rad]` buffers (utils/pose_emb.py:7-56, utils/positional_emb.py:6-54).
The sinusoids themselves are evaluated inside the HIP kernels (tbx_knn_embed / tbx_agent_prep / tbx_pose_embed)."""
import torch
from torch import Tensor, nn

from .. import hip


class PositionalEmbedding(nn.Module):
    def __init__(self, dim: int, theta: float = 10000):
        super().__init__()
        assert dim % 2 == 0
        self.dim = dim
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.register_buffer("freqs", freqs.repeat_interleave(2, 0))


class PositionalEmbeddingRad(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        assert dim % 2 == 0
        self.dim = dim
        self.register_buffer("freqs", (torch.arange(0, dim // 2) + 1.0).repeat_interleave(2, 0))


class PoseEmb(nn.Module):
    def __init__(self, mode: str, pe_dim: int = 256, theta_xy: float = 1e3, theta_cs: float = 1e1):
        super().__init__()
        self.mode = mode
        if mode == "mpa_pl":
            self.out_dim = 7  # evaluated by tbx_map_prep
        elif mode == "pe_xy_yaw":
            self.out_dim = pe_dim
            self.pe_xy = PositionalEmbedding(dim=pe_dim // 4, theta=theta_xy)
            self.pe_yaw = PositionalEmbeddingRad(dim=pe_dim // 2)
        else:
            raise NotImplementedError(f"PoseEmb mode {mode} is not on the default hot path")

    def forward(self, xy: Tensor, dir: Tensor) -> Tensor:
        if self.mode != "pe_xy_yaw" or dir.shape[-1] != 1:
            raise NotImplementedError("stand-alone PoseEmb.forward supports pe_xy_yaw with yaw input")
        pose3 = torch.cat([xy, dir], -1).reshape(-1, 3).contiguous().float()
        out = hip.pose_embed(pose3, self.pe_xy.freqs, self.pe_yaw.freqs, self.out_dim)
        return out.view(*xy.shape[:-1], self.out_dim)
