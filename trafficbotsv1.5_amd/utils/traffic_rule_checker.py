"""`TrafficRuleChecker` (utils/traffic_rule_checker.py:10-85) as the holder of the static tensors the rule checks
need. The two checks that feed back into the closed loop (outside-map -> disable agent, destination reached ->
disable navi) run inside `tbx_sim_step`; the metric-only checks (collision, road edge, red light, passive) are the
next row of the scope table and are not computed yet."""
from typing import Optional

from torch import Tensor


class TrafficRuleChecker:
    def __init__(self, mp_boundary: Tensor, mp_valid: Tensor, mp_type: Tensor, mp_pos: Tensor, mp_dir: Tensor, ag_type: Tensor,
                 ag_size: Tensor, ag_goal: Optional[Tensor], ag_dest: Optional[Tensor], tl_valid: Tensor, tl_pose: Tensor,
                 disable_check: bool, collision_size_scale: float = 1.1) -> None:
        self.mp_boundary, self.mp_valid, self.mp_type = mp_boundary, mp_valid, mp_type
        self.mp_pos, self.mp_dir = mp_pos, mp_dir
        self.ag_type, self.ag_size, self.ag_goal, self.ag_dest = ag_type, ag_size, ag_goal, ag_dest
        self.tl_valid, self.tl_pose, self.disable_check = tl_valid, tl_pose, disable_check
