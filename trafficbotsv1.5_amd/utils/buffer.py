"""`RolloutBuffer` (utils/buffer.py:7-146) as a view over the preallocated [n_sc, n_ag, n_step, ...] device logs that
`tbx_sim_step` writes in place (the reference appends to Python lists and stacks at the end)."""
from typing import Dict, Optional

from torch import Tensor


class RolloutBuffer:
    def __init__(self, step_end: int, step_current: int) -> None:
        self.step_start, self.step_end, self.step_future_start = 1, step_end, step_current
        self.pred_valid: Optional[Tensor] = None   # [n_sc, n_ag, n_step] bool, validity before the step's override
        self.pred_pose: Optional[Tensor] = None    # [n_sc, n_ag, n_step, 3]
        self.pred_motion: Optional[Tensor] = None  # [n_sc, n_ag, n_step, 3]
        self.violation: Dict[str, Tensor] = {}     # outside_map / dest_reached [n_sc, n_ag, n_step] bool
        self.vis_dict: Dict[str, Tensor] = {}      # action [n_sc, n_ag, n_step, 2], tl_state [n_sc, n_tl, n_step, 5]
        self.log_prob = None

    def flatten_joint_future(self, n_joint_future: int) -> None:
        def split(t: Tensor) -> Tensor:
            return t.view(t.shape[0] // n_joint_future, n_joint_future, *t.shape[1:])

        self.pred_valid, self.pred_pose, self.pred_motion = split(self.pred_valid), split(self.pred_pose), split(self.pred_motion)
        self.violation = {k: split(v) for k, v in self.violation.items()}
        self.vis_dict = {k: split(v) for k, v in self.vis_dict.items()}
