"""`TrainingMetrics` with the reference's constructor / `update` / `compute` / `reset` (models/metrics/training.py:11-189), for the
default switches (no relevance weighting, no temporal discount). `WaymoMotion.training_step` computes the same loss inside the
captured step (train_graph.training_loss: one fused expression, no host branch); this class is the reference-shaped accumulator for
callers that drive `update(buffer, ...)` themselves - same numbers (tests/test_hip_utils_surface.py)."""
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .loss import BalancedKL


class TrainingMetrics(nn.Module):
    def __init__(self, prefix: str, train_navi: bool, train_latent: bool, w_vae_kl: float, kl_balance_scale: float, kl_free_nats: float,
                 kl_for_unseen_agent: bool, w_diffbar_reward: float, w_navi: float, w_tl_state: float, w_relevant_agent: float,
                 p_loss_for_irrelevant: float, step_training_start: int, temporal_discount: float = -1.0,
                 loss_for_teacher_forcing: bool = False) -> None:
        super().__init__()
        if w_relevant_agent > 0 or p_loss_for_irrelevant < 1.0 or temporal_discount > 0:
            raise NotImplementedError("relevance weighting / temporal discount are off in the default configuration")
        self.prefix, self.step_training_start, self.loss_for_teacher_forcing = prefix, step_training_start, loss_for_teacher_forcing
        self.train_latent, self.train_navi = train_latent, train_navi
        self.w_vae_kl, self.kl_for_unseen_agent = w_vae_kl, kl_for_unseen_agent
        self.l_vae_kl = BalancedKL(kl_balance_scale=kl_balance_scale, kl_free_nats=kl_free_nats)
        self.w_diffbar_reward, self.use_diffbar_reward = w_diffbar_reward, w_diffbar_reward > 0
        self.w_navi, self.w_tl_state, self.train_tl_state = w_navi, w_tl_state, w_tl_state > 0
        self._names = ("vae_kl_counter", "vae_kl", "diffbar_reward_counter", "diffbar_reward", "dr_il_pos", "dr_il_rot", "dr_il_spd",
                       "dr_rule_apx", "navi_loss", "navi_counter", "tl_state_loss", "tl_state_counter")
        for n in self._names:
            self.register_buffer(n, torch.tensor(0.0), persistent=False)

    def reset(self) -> None:
        for n in self._names:
            getattr(self, n).zero_()

    def forward(self, *a, **kw) -> Dict[str, Tensor]:  # torchmetrics' Metric.forward: update + this batch's value
        self.reset()
        self.update(*a, **kw)
        return self.compute()

    def update(self, buffer, ag_role: Tensor, navi_pred, navi_gt: Tensor, latent_post, latent_prior) -> None:
        dev = buffer.pred_valid.device
        if self.vae_kl.device != dev:
            self.to(dev)
        with torch.no_grad():
            lv = buffer.pred_valid.clone()
            if self.step_training_start > 0:
                lv[:, :, : self.step_training_start] &= False
            if not self.loss_for_teacher_forcing:
                lv &= ~buffer.mask_teacher_forcing
            any_valid = lv.any(-1)
        if self.train_latent:
            kv = (latent_post.valid if self.kl_for_unseen_agent else latent_prior.valid) & any_valid
            err = self.l_vae_kl.compute(latent_post.distribution, latent_prior.distribution)
            self.vae_kl_counter += kv.sum()
            self.vae_kl = self.vae_kl + err.masked_fill(~kv, 0.0).sum()
        if self.use_diffbar_reward:
            rv = lv & buffer.diffbar_reward["diffbar_reward_valid"]
            self.diffbar_reward = self.diffbar_reward + buffer.diffbar_reward["diffbar_reward"].masked_fill(~rv, 0.0).sum()
            self.diffbar_reward_counter += rv.sum()
            if "r_imitation_pos" in buffer.diffbar_reward:
                self.dr_il_pos += buffer.diffbar_reward["r_imitation_pos"].sum()
                self.dr_il_rot += buffer.diffbar_reward["r_imitation_rot"].sum()
                self.dr_il_spd += buffer.diffbar_reward["r_imitation_spd"].sum()
        if self.train_navi:
            nv = navi_pred.valid & any_valid
            self.navi_loss = self.navi_loss + (-navi_pred.log_prob(navi_gt)).masked_fill(~nv, 0).sum()
            self.navi_counter += nv.sum()
        if self.train_tl_state:
            inv = buffer.tl_state_nll_invalid
            self.tl_state_loss = self.tl_state_loss + buffer.tl_state_nll.masked_fill(inv, 0.0).sum()
            self.tl_state_counter += (~inv).sum()

    def compute(self) -> Dict[str, Tensor]:
        p, out = self.prefix, {}
        loss = 0.0
        if self.train_latent and self.vae_kl_counter > 0:
            out[f"{p}/vae_kl"] = self.w_vae_kl * self.vae_kl / self.vae_kl_counter
            loss = loss + out[f"{p}/vae_kl"]
        if self.use_diffbar_reward and self.diffbar_reward_counter > 0:
            out[f"{p}/diffbar_reward"] = self.w_diffbar_reward * self.diffbar_reward / self.diffbar_reward_counter
            for k in ("dr_il_pos", "dr_il_rot", "dr_il_spd", "dr_rule_apx"):
                out[f"{p}/{k}"] = getattr(self, k) / self.diffbar_reward_counter
            loss = loss - out[f"{p}/diffbar_reward"]
        if self.train_navi and self.navi_counter > 0:
            out[f"{p}/navi_loss"] = self.w_navi * self.navi_loss / self.navi_counter
            loss = loss + out[f"{p}/navi_loss"]
        if self.train_tl_state and self.tl_state_counter > 0:
            out[f"{p}/tl_state_loss"] = self.w_tl_state * self.tl_state_loss / self.tl_state_counter
            loss = loss + out[f"{p}/tl_state_loss"]
        out[f"{p}/loss"] = loss
        return out
