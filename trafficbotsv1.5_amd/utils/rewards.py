"""`DifferentiableReward` with the reference's constructor and `get` (utils/rewards.py:9-85), default configuration: the three
imitation terms (SmoothL1 position / speed, "cosine" heading - models/metrics/loss.py:9-36), `w_collision = 0`. One step per call
through tbx_diffbar_reward - the expressions tbx_sim_step logs for every step of a rollout (`RolloutBuffer.diffbar_reward`) and
tbx_train_chain_fwd / _bwd differentiates inside the training step; this stand-alone call is forward-only."""
from typing import Dict, Optional

import torch
from torch import Tensor

from .. import hip


class DifferentiableReward:
    def __init__(self, l_pos, l_rot, l_spd, w_collision: float, use_il_loss: bool, reduce_collsion_with_max: bool, is_enabled: bool = True):
        self.w_collision, self.reduce_collsion_with_max, self.is_enabled, self.use_il_loss = w_collision, reduce_collsion_with_max, is_enabled, use_il_loss
        if w_collision > 0:
            raise NotImplementedError("the relaxed-collision term (w_collision > 0) is off in the default configuration (sim_agent.yaml:189)")
        if use_il_loss:
            if not (l_pos["criterion"] == l_spd["criterion"] == "SmoothL1Loss" and l_rot.get("angular_type") == "cosine"):
                raise NotImplementedError("tbx_diffbar_reward implements the default criteria: SmoothL1Loss position / speed, cosine heading")
            self.il_w_pos, self.il_w_rot, self.il_w_spd = float(l_pos["weight"]), float(l_rot["weight"]), float(l_spd["weight"])

    @torch.no_grad()
    def get(self, pred_valid: Tensor, pred_pose: Tensor, pred_motion: Tensor, gt_valid: Optional[Tensor], gt_pose: Optional[Tensor],
            gt_motion: Optional[Tensor], ag_size: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """[n_sc, n_ag] / [n_sc, n_ag, 3] tensors of ONE step -> the reference's reward dict."""
        if not self.is_enabled:
            return {}
        c = lambda t: t.float().contiguous()
        u8 = lambda t: t.to(torch.uint8).contiguous()
        il = self.use_il_loss and gt_valid is not None
        w = (self.il_w_pos, self.il_w_rot, self.il_w_spd) if il else (0.0, 0.0, 0.0)
        out4, valid = hip.diffbar_reward(u8(pred_valid), c(pred_pose), c(pred_motion), u8(gt_valid) if il else None,
                                         c(gt_pose) if il else None, c(gt_motion) if il else None, *w)
        return {"diffbar_reward_valid": valid.bool(), "diffbar_reward": out4[..., 3], "r_imitation_pos": out4[..., 0],
                "r_imitation_rot": out4[..., 1], "r_imitation_spd": out4[..., 2], "r_traffic_rule_approx": torch.zeros_like(out4[..., 0])}
