"""`LatentEncoder` / `DistEncoder` (models/latent_encoder.py:14-253): CVAE posterior over the down-sampled ground
truth episode (window 19) with its own tl / agent encoders; the prior is a frozen standard normal (`std_gaus`)."""
from copy import deepcopy
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from ..engine import kv_tables
from ..utils.pose_emb import PoseEmb
from .agent_encoder import AgentEncoder
from .modules.distributions import DiagGaussian, MyDist
from .modules.mlp import MLP
from .traffic_light import TrafficLightEncoder


class DistEncoder(nn.Module):
    def __init__(self, hidden_dim: int, out_dim: int, branch_type: bool, dist_type: str, mlp_use_layernorm: bool,
                 log_std: Optional[float], n_cat: int, n_layer: int) -> None:
        super().__init__()
        self.dist_type, self.branch_type = dist_type, branch_type
        if dist_type == "std_gaus":
            self.skip_forward = True
            self.mean = nn.Parameter(torch.zeros(1, 1, out_dim), requires_grad=False)
            self.log_std = nn.Parameter(torch.zeros(out_dim), requires_grad=False)
        elif dist_type == "diag_gaus" and not branch_type and log_std is not None:
            self.skip_forward = False
            self.mlp_mean = MLP([hidden_dim] * n_layer + [out_dim], end_layer_activation=False, use_layernorm=mlp_use_layernorm)
            self.log_std = nn.Parameter(log_std * torch.ones(out_dim), requires_grad=True)
        else:
            raise NotImplementedError("default config: diag_gaus posterior (fixed log_std), std_gaus prior")

    def forward(self, x: Tensor, valid: Tensor, ag_type: Tensor) -> MyDist:
        if self.dist_type == "std_gaus":
            return DiagGaussian(self.mean.expand(*valid.shape, -1), self.log_std, valid=valid)
        return DiagGaussian(self.mlp_mean(x, ~valid), self.log_std, valid=valid)


class LatentEncoder(nn.Module):
    def __init__(self, latent_dim: int, temporal_down_sample_rate: int, share_post_prior_encoders: bool, latent_prior,
                 latent_post, tl_encoder, ag_encoder, pose_rpe: Optional[PoseEmb], time_step_gt: int):
        super().__init__()
        self.out_dim, self.dummy = latent_dim, latent_dim <= 0
        self.temporal_down_sample_rate = temporal_down_sample_rate
        if self.dummy:
            return
        r = temporal_down_sample_rate
        win = (time_step_gt + 1) // r + 1 if r > 1 else time_step_gt + 1
        tl_encoder, ag_encoder = deepcopy(tl_encoder), deepcopy(ag_encoder)
        tl_encoder["temp_window_size"] = win
        ag_encoder["temp_window_size"] = win
        self.tl_encoder_post = TrafficLightEncoder(pose_rpe=pose_rpe, **tl_encoder)
        self.ag_encoder_post = AgentEncoder(pose_rpe=pose_rpe, **ag_encoder)
        if share_post_prior_encoders:
            self.tl_encoder_prior, self.ag_encoder_prior = self.tl_encoder_post, self.ag_encoder_post
        else:  # kept for state-dict parity; dead under the std_gaus prior (SURVEY.md finding 3)
            self.tl_encoder_prior = TrafficLightEncoder(pose_rpe=pose_rpe, **tl_encoder)
            self.ag_encoder_prior = AgentEncoder(pose_rpe=pose_rpe, **ag_encoder)
        self.latent_dist_prior = DistEncoder(ag_encoder["hidden_dim"], latent_dim, **latent_prior)
        self.latent_dist_post = DistEncoder(ag_encoder["hidden_dim"], latent_dim, **latent_post)

    def forward(self, ag_valid: Tensor, ag_attr: Tensor, ag_motion: Tensor, ag_pose: Tensor, ag_type: Tensor,
                tl_state: Tensor, mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor], posterior: bool) -> Optional[MyDist]:
        if self.dummy:
            return None
        dist = self.latent_dist_post if posterior else self.latent_dist_prior
        if dist.skip_forward:
            return dist(ag_attr, ag_valid.any(-1), ag_type)
        r = self.temporal_down_sample_rate
        if r > 1:
            assert (ag_valid.shape[-1] - 1) % r == 0
            ag_valid, ag_motion, ag_pose, tl_state = ag_valid[:, :, ::r], ag_motion[:, :, ::r], ag_pose[:, :, ::r], tl_state[:, :, ::r]
        tl_enc = self.tl_encoder_post if posterior else self.tl_encoder_prior
        ag_enc = self.ag_encoder_post if posterior else self.ag_encoder_prior
        n, A, _ = ag_valid.shape
        L = tl_state.shape[1]
        hist_tl = tl_enc.states_to_hist(tl_state, tl_enc.temp_window_size)
        tl_feat = tl_enc.encode(hist_tl, tl_tokens)
        tl_kv = kv_tables(tl_feat, ag_enc.tl_kv_layers())
        hv, hp, hm = ag_enc.pad_hist(ag_valid, ag_pose, ag_motion, ag_enc.temp_window_size)
        feat, _ = ag_enc.encode(hv, hp, hm, ag_attr.float().contiguous(), mp_tokens, tl_tokens["tl_token_invalid_u8"],
                                tl_tokens["tl_token_pose"], tl_kv, mp_batch_div=tl_tokens.get("mp_batch_div", 1))
        return dist(feat.view(n, A, -1), ag_valid.any(-1), ag_type)
