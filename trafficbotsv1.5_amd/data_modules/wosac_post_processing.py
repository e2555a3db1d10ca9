"""`WOSACPostProcessing._filter_futures` (data_modules/wosac_post_processing.py:12-64 of the reference): of the K joint
futures simulated per scene keep the 32 with the fewest collisions / road-edge crossings among the agents that carry a
role. The step right after the rollout and its rule checks; scoring, ranking and the gather of the
kept trajectories are one C-ABI call (`tbx_filter_futures`) on the device-resident rollout log.

The rest of the reference class (scenario-frame -> global-frame transform, protobuf submission) is out of scope.
"""
import torch
from torch import Tensor, nn

from .. import hip
from ..utils.buffer import RolloutBuffer


class WOSACPostProcessing(nn.Module):
    def __init__(self, step_gt: int, step_current: int, const_vel_z_sim: bool, const_vel_no_sim: bool, w_road_edge: float,
                 use_wosac_col: bool) -> None:
        super().__init__()
        self.step_gt, self.step_current = step_gt, step_current
        self.const_vel_z_sim, self.const_vel_no_sim = const_vel_z_sim, const_vel_no_sim
        self.n_joint_future = 32  # from the WOSAC challenge (wosac_post_processing.py:28)
        self.w_road_edge, self.use_wosac_col = w_road_edge, use_wosac_col
        self.last_idx = None   # [n_sc, 32] i32 rollouts kept by the last call, ascending (score, index)
        self.last_score = None  # [n_sc, K] f32

    @torch.no_grad()
    def _filter_futures(self, buffer: RolloutBuffer, ag_role: Tensor) -> Tensor:
        """buffer.pred_pose [n_sc, K, A, T, 3], buffer.violation[*] [n_sc, K, A, T] bool, ag_role [n_sc, A, 3] bool
        -> trajs [n_sc, min(K, 32), A, T - step_future_start, 3]. Ties between equally bad rollouts go to the lower index
        (the reference's topk leaves them unspecified)."""
        start = buffer.step_future_start
        n_sc, K, A, T = buffer.pred_pose.shape[:4]
        if K <= self.n_joint_future:
            return buffer.pred_pose[:, :, :, start:]
        col = buffer.violation["collided_wosac" if self.use_wosac_col else "collided"]
        bit = hip.RULE_COLLIDED_WOSAC if self.use_wosac_col else hip.RULE_COLLIDED
        flags = (col.to(torch.uint8) * bit + buffer.violation["run_road_edge"].to(torch.uint8) * hip.RULE_RUN_ROAD_EDGE)
        self.last_score, self.last_idx, trajs = hip.filter_futures(
            flags.reshape(n_sc * K, A, T).contiguous(), bit, ag_role.any(-1).to(torch.uint8).contiguous(), n_sc, K, start,
            float(self.w_road_edge), self.n_joint_future, pred_pose=buffer.pred_pose.reshape(n_sc * K, A, T, 3).float().contiguous())
        return trajs

    def forward(self, batch, buffer: RolloutBuffer):
        raise NotImplementedError("global-frame transform + WOSAC submission records are outside the hot path (SURVEY.md §8f)")
