"""`TeacherForcing` (utils/teacher_forcing.py:8-168): which agents are overridden by ground truth at which step.
The whole [n_sc, n_ag, n_step] mask is built once per rollout on the device; `tbx_sim_step` indexes it by the device
step counter, so there is no per-step host work."""
from typing import Dict, Tuple

import torch
from torch import Tensor


class TeacherForcing:
    def __init__(self, step_spawn_agent: int = 10, step_warm_start: int = 10, step_horizon: int = 0,
                 step_horizon_decrease_per_epoch: int = 0, prob_forcing_agent: float = 0,
                 prob_forcing_agent_decrease_per_epoch: float = 0, prob_scheduled_sampling: float = 0,
                 prob_scheduled_sampling_decrease_per_epoch: float = 0, gt_sdc: bool = False, threshold_xy: float = -1.0,
                 threshold_yaw: float = -1.0, threshold_spd: float = -1.0) -> None:
        if threshold_xy > 0 or threshold_yaw > 0 or threshold_spd > 0:
            raise NotImplementedError("error-threshold resets are off in every default schedule")
        self.step_spawn_agent, self.step_warm_start = step_spawn_agent, step_warm_start
        self.step_horizon, self.step_horizon_decrease_per_epoch = step_horizon, step_horizon_decrease_per_epoch
        self.prob_forcing_agent, self.prob_forcing_agent_decrease_per_epoch = prob_forcing_agent, prob_forcing_agent_decrease_per_epoch
        self.prob_scheduled_sampling = prob_scheduled_sampling
        self.prob_scheduled_sampling_decrease_per_epoch = prob_scheduled_sampling_decrease_per_epoch
        self.gt_sdc = gt_sdc

    @torch.no_grad()
    def init(self, ag_valid: Tensor, ag_pose: Tensor, ag_motion: Tensor, tl_state: Tensor, current_epoch: int) -> None:
        self.ag_valid, self.ag_pose, self.ag_motion, self.tl_state = ag_valid, ag_pose, ag_motion, tl_state
        self.tl_teacher_forcing = torch.ones_like(tl_state[..., 0], dtype=torch.bool)
        tf = torch.zeros_like(ag_valid)
        tf[:, :, 0] |= ag_valid[:, :, 0]
        if self.step_spawn_agent > 0:
            spawn = (~ag_valid[:, :, :-1]) & ag_valid[:, :, 1:]
            spawn[:, :, self.step_spawn_agent:] = False
            tf[:, :, 1:] |= spawn
        if self.step_warm_start >= 0:
            tf[:, :, : self.step_warm_start + 1] |= ag_valid[:, :, : self.step_warm_start + 1]
        horizon = int(self.step_horizon - self.step_horizon_decrease_per_epoch * current_epoch)
        if horizon > 0:
            tf[:, :, :horizon] |= ag_valid[:, :, :horizon]
        p_agent = self.prob_forcing_agent - self.prob_forcing_agent_decrease_per_epoch * current_epoch
        if p_agent > 0:
            pick = torch.bernoulli(torch.full_like(ag_valid[:, :, 0], p_agent, dtype=torch.float32)).bool()
            tf |= pick.unsqueeze(-1) & ag_valid
        p_ss = self.prob_scheduled_sampling - self.prob_scheduled_sampling_decrease_per_epoch * current_epoch
        if p_ss > 0:
            tf |= torch.bernoulli(torch.full_like(ag_valid, p_ss, dtype=torch.float32)).bool() & ag_valid
        if self.gt_sdc:
            tf[:, 0] |= ag_valid[:, 0]
        self.ag_teacher_forcing = tf

    @torch.no_grad()
    def get(self, step: int, pred_valid: Tensor, pred_pose: Tensor, pred_motion: Tensor) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
        if 0 < step < self.ag_teacher_forcing.shape[-1]:
            ag = {"valid": self.ag_teacher_forcing[:, :, step], "pose": self.ag_pose[:, :, step], "motion": self.ag_motion[:, :, step]}
        else:
            ag = {"valid": torch.zeros_like(self.ag_teacher_forcing[:, :, 0]), "pose": torch.zeros_like(self.ag_pose[:, :, 0]),
                  "motion": torch.zeros_like(self.ag_motion[:, :, 0])}
        if 0 < step < self.tl_teacher_forcing.shape[-1]:
            tl = {"valid": self.tl_teacher_forcing[:, :, step], "state": self.tl_state[:, :, step]}
        else:
            tl = {"valid": torch.zeros_like(self.tl_teacher_forcing[:, :, 0]), "state": torch.zeros_like(self.tl_state[:, :, 0])}
        return ag, tl
