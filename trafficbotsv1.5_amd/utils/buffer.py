"""`RolloutBuffer` (utils/buffer.py:7-146) with the reference's fields.

Two ways to fill it:
  * the rollout engine (utils/rollout_engine.py) hands over views of the preallocated [n_sc, n_ag, n_step, ...] device logs that
    `tbx_sim_step` wrote in place (`RolloutEngine.buffer`) - already stacked, `finish()` is then a no-op;
  * a step-wise driver (`WaymoMotion.rollout(..., player_policy=...)`, the reference's Python loop) calls `add` per step and
    `finish` at the end, exactly like the reference (lists, stacked along dim 2).
"""
from typing import Dict, List, Optional

import torch
from torch import Tensor


class RolloutBuffer:
    def __init__(self, step_end: int, step_current: int) -> None:
        self.step_start, self.step_end, self.step_future_start = 1, step_end, step_current
        self.pred_valid = []    # [n_sc, n_ag, n_step] bool, validity before the step's override
        self.pred_pose = []     # [n_sc, n_ag, n_step, 3]
        self.pred_motion = []   # [n_sc, n_ag, n_step, 3]
        self.action_log_prob = []       # [n_sc, n_ag, n_step]
        self.navi_log_prob: List[Tensor] = []        # -> [n_sc, n_ag, n_step_navi_update]
        self.navi_log_prob_valid: List[Tensor] = []  # -> [n_sc, n_ag, n_step_navi_update]
        self.tl_state_nll = []          # [n_sc, n_tl, n_step]
        self.tl_state_nll_invalid = []  # [n_sc, n_tl, n_step]
        self.diffbar_reward: Dict[str, Tensor] = {}  # diffbar_reward(_valid), r_imitation_pos / rot / spd [n_sc, n_ag, n_step]
        self.mask_teacher_forcing = []  # [n_sc, n_ag, n_step] bool
        self.violation: Dict[str, Tensor] = {}  # [n_sc, n_ag, n_step] bool
        self.vis_dict: Dict[str, Tensor] = {}   # action [n_sc, n_ag, n_step, 2], tl_state [n_sc, n_tl, n_step, 5]
        self.log_prob = None
        self._finished = False

    # ------------------------------------------------------------------ the reference's per-step interface (buffer.py:39-78)
    def add(self, violation: Dict[str, Tensor], diffbar_reward: Dict[str, Tensor], tl_state_nll: Tensor, tl_state_nll_invalid: Tensor,
            vis_dict: Dict[str, Tensor], pred_valid: Tensor, pred_pose: Tensor, pred_motion: Tensor, action_log_prob: Tensor,
            ag_override: Dict[str, Tensor], **kwargs) -> None:
        self.pred_valid.append(pred_valid)
        self.pred_pose.append(pred_pose)
        self.pred_motion.append(pred_motion)
        self.tl_state_nll.append(tl_state_nll)
        self.tl_state_nll_invalid.append(tl_state_nll_invalid)
        for name, src in (("violation", violation), ("diffbar_reward", diffbar_reward)):
            dst = getattr(self, name)
            if len(dst) == 0:
                dst.update({k: [] for k in src})
            for k, v in src.items():
                dst[k].append(v)
        self.action_log_prob.append(action_log_prob)
        if len(self.vis_dict) == 0:
            self.vis_dict = {k: [] for k in vis_dict}
        for k, v in vis_dict.items():
            if v is not None:
                self.vis_dict[k].append(v)
        self.mask_teacher_forcing.append(ag_override["valid"])

    def finish(self) -> None:
        """buffer.py:80-104: lists -> tensors stacked along dim 2. No-op for a buffer the engine filled."""
        if self._finished:
            return
        st = lambda l: torch.stack(l, dim=2)
        self.pred_valid, self.pred_pose, self.pred_motion = st(self.pred_valid), st(self.pred_pose), st(self.pred_motion)
        self.tl_state_nll, self.tl_state_nll_invalid = st(self.tl_state_nll), st(self.tl_state_nll_invalid)
        self.navi_log_prob, self.navi_log_prob_valid = st(self.navi_log_prob), st(self.navi_log_prob_valid)
        self.violation = {k: st(v) for k, v in self.violation.items()}
        self.diffbar_reward = {k: st(v) for k, v in self.diffbar_reward.items()}
        self.action_log_prob = st(self.action_log_prob)
        self.vis_dict = {k: (st(v) if len(v) > 0 else v) for k, v in self.vis_dict.items()}
        self.mask_teacher_forcing = st(self.mask_teacher_forcing)
        self._finished = True

    def add_navi_log_prob(self, ag_navi_log_prob: Tensor, mask_navi_reached: Tensor) -> None:
        self.navi_log_prob.append(ag_navi_log_prob)
        self.navi_log_prob_valid.append(mask_navi_reached)

    def compute_log_prob(self, latent_log_prob: Optional[Tensor]) -> None:
        """buffer.py:110-117."""
        self.log_prob = (self.navi_log_prob * self.navi_log_prob_valid).sum(-1)
        self.log_prob = self.log_prob / self.navi_log_prob_valid.sum(-1)
        self.log_prob = self.log_prob.masked_fill(~self.navi_log_prob_valid.any(-1), 0)
        if latent_log_prob is not None:
            self.log_prob = self.log_prob + latent_log_prob.view(self.log_prob.shape)

    def flatten_joint_future(self, n_joint_future: int) -> None:
        """buffer.py:119-146: [n_sc * K, ...] -> [n_sc, K, ...] for every field."""
        def split(t):
            return t.view(t.shape[0] // n_joint_future, n_joint_future, *t.shape[1:]) if torch.is_tensor(t) else t

        for name in ("pred_valid", "pred_pose", "pred_motion", "action_log_prob", "navi_log_prob", "navi_log_prob_valid",
                     "tl_state_nll", "tl_state_nll_invalid", "mask_teacher_forcing"):
            setattr(self, name, split(getattr(self, name)))
        self.violation = {k: split(v) for k, v in self.violation.items()}
        self.diffbar_reward = {k: split(v) for k, v in self.diffbar_reward.items()}
        self.vis_dict = {k: split(v) for k, v in self.vis_dict.items()}
