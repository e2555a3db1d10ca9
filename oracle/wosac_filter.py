"""ORACLE (test infrastructure, never on the product path): CPU restatement of the WOSAC rollout filter, SURVEY.md §8f
row 3 - `WOSACPostProcessing._filter_futures` (data_modules/wosac_post_processing.py:31-64 of the reference): of the K
simulated joint futures of a scene keep the 32 with the fewest rule violations.

Pinned: tests/test_oracle_rules.py checks it against tests/golden/filter.npz (outputs of the reference's own
`_filter_futures` on seeded inputs, tests/golden/make_golden.py: gen_filter)."""
from typing import Tuple

import torch
from torch import Tensor


def rollout_scores(collided: Tensor, run_road_edge: Tensor, ag_role: Tensor, step_future_start: int, w_road_edge: float) -> Tensor:
    """wosac_post_processing.py:47-57. collided / run_road_edge [n_sc, K, A, T] bool, ag_role [n_sc, A, 3] bool -> [n_sc, K]."""
    role = (ag_role.any(-1) * 1.0).unsqueeze(1)
    col = (collided[..., step_future_start:].any(-1) * role).sum(-1)
    edge = (run_road_edge[..., step_future_start:].any(-1) * role).sum(-1)
    return col + edge * w_road_edge


def filter_futures(pred_pose: Tensor, collided: Tensor, run_road_edge: Tensor, ag_role: Tensor, step_future_start: int,
                   w_road_edge: float, n_joint_future: int = 32) -> Tuple[Tensor, Tensor]:
    """wosac_post_processing.py:31-64 -> (trajs [n_sc, min(K, 32), A, T - start, 3], idx [n_sc, 32] or None)."""
    trajs = pred_pose[:, :, :, step_future_start:]
    if trajs.shape[1] <= n_joint_future:
        return trajs, None
    score = rollout_scores(collided, run_road_edge, ag_role, step_future_start, w_road_edge)
    _, idx = torch.topk(score, n_joint_future, dim=-1, largest=False, sorted=False)
    return trajs[torch.arange(trajs.shape[0]).unsqueeze(1), idx], idx
