"""Seeded synthetic fixed-shape scenes in the packed-h5 batch layout of the reference.

Keys / dtypes / shapes follow `src/data_modules/data_h5_womd.py:102-134` (training episode, 91 steps).
The generator itself is ours: there is no dataset on the box, and fixed-shape synthetic
scenes are what BASELINE.json's configs are quoted on.
"""
import math
from typing import Dict

import torch

# agent type -> destination polyline types allowed by NaviPredictor's logits mask (reference navigation.py:263-277)
_DEST_TYPES = {0: (0, 1, 2, 4), 1: (4,), 2: (3, 4)}


def make_scene(
    n_sc: int = 1, n_ag: int = 64, n_mp: int = 1024, n_tl: int = 128, n_step: int = 91, seed: int = 0,
    n_pl_node: int = 20, ragged: bool = True,
) -> Dict[str, torch.Tensor]:
    """Returns a CPU batch dict (fp32 / bool / int64) of `n_sc` scenes, scene i seeded with `seed + i`."""
    scenes = [_one_scene(n_ag, n_mp, n_tl, n_step, seed + i, n_pl_node, ragged) for i in range(n_sc)]
    return {k: torch.stack([s[k] for s in scenes], 0) for k in scenes[0]}


def _one_scene(n_ag, n_mp, n_tl, n_step, seed, n_node, ragged) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g)
    out: Dict[str, torch.Tensor] = {}

    # ---- map polylines: 1 m spacing, slowly curving
    start = (U(n_mp, 2) - 0.5) * 300.0
    head0 = (U(n_mp) - 0.5) * 2 * math.pi
    dh = torch.randn(n_mp, n_node, generator=g) * 0.02
    dh[:, 0] = 0
    head = head0[:, None] + dh.cumsum(1)  # [n_mp, n_node]
    seg = torch.stack([head.cos(), head.sin()], -1)  # unit segments
    pos = start[:, None, :] + torch.cat([torch.zeros(n_mp, 1, 2), seg[:, :-1].cumsum(1)], 1)
    n_valid = torch.randint(1, n_node + 1, (n_mp,), generator=g)
    mp_valid = torch.arange(n_node)[None, :] < n_valid[:, None]
    mp_type_idx = torch.randint(0, 11, (n_mp,), generator=g)
    # make sure every destination class exists
    for t in range(5):
        mp_type_idx[t % n_mp] = t
    mp_valid[: min(5, n_mp), 0] = True
    out["map/valid"] = mp_valid
    out["map/type"] = torch.nn.functional.one_hot(mp_type_idx, 11).bool()
    out["map/pos"] = torch.cat([pos, torch.zeros(n_mp, n_node, 1)], -1)
    out["map/dir"] = torch.cat([seg, torch.zeros(n_mp, n_node, 1)], -1)
    out["map/boundary"] = torch.tensor([-200.0, 200.0, -200.0, 200.0])

    # ---- traffic lights on distinct lanes
    tl_idx = torch.randperm(n_mp, generator=g)[:n_tl]
    tl_on = U(n_tl) < 0.5
    tl_on[0] = True
    out["tl_lane/idx"] = tl_idx
    out["tl_lane/valid"] = tl_on[:, None].expand(-1, n_step).clone()
    n_blk = (n_step + 9) // 10
    st = torch.randint(0, 5, (n_tl, n_blk), generator=g).repeat_interleave(10, 1)[:, :n_step]
    out["tl_lane/state"] = torch.nn.functional.one_hot(st, 5).bool() & out["tl_lane/valid"][..., None]
    # stop-point lights are only sized (tl_mode=lane): keep the reference's 50
    out["tl_stop/valid"] = torch.zeros(50, n_step, dtype=torch.bool)
    out["tl_stop/state"] = torch.zeros(50, n_step, 5, dtype=torch.bool)
    out["tl_stop/pos"] = torch.zeros(50, 3)
    out["tl_stop/dir"] = torch.zeros(50, 3)

    # ---- agents: constant speed, constant yaw rate
    xy0 = (U(n_ag, 2) - 0.5) * 150.0
    yaw0 = (U(n_ag) - 0.5) * 2 * math.pi
    spd = U(n_ag) * 15.0
    yaw_rate = (U(n_ag) - 0.5) * 0.2
    t = torch.arange(n_step, dtype=torch.float32) * 0.1
    yaw = yaw0[:, None] + yaw_rate[:, None] * t[None, :]
    vel = torch.stack([spd[:, None] * yaw.cos(), spd[:, None] * yaw.sin()], -1)  # [n_ag, n_step, 2]
    xy = xy0[:, None, :] + torch.cat([torch.zeros(n_ag, 1, 2), (vel[:, :-1] * 0.1).cumsum(1)], 1)
    valid = torch.ones(n_ag, n_step, dtype=torch.bool)
    if ragged and n_ag >= 8:
        # a few late spawns, a few early exits, one agent never observed
        k = max(1, n_ag // 8)
        late = torch.randint(1, 9, (k,), generator=g)
        for i in range(k):
            valid[1 + i, : int(late[i])] = False
        for i in range(k):
            valid[1 + k + i, 30 + 5 * i :] = False
        valid[n_ag - 1] = False
    ag_type = torch.randint(0, 3, (n_ag,), generator=g)
    ag_type[0] = 0
    out["agent/valid"] = valid
    out["agent/pos"] = torch.cat([xy, torch.zeros(n_ag, n_step, 1)], -1)
    out["agent/vel"] = vel
    out["agent/spd"] = spd[:, None, None].expand(-1, n_step, 1).clone()
    out["agent/acc"] = torch.zeros(n_ag, n_step, 1)
    out["agent/yaw_bbox"] = yaw[..., None].clone()
    out["agent/yaw_rate"] = yaw_rate[:, None, None].expand(-1, n_step, 1).clone()
    out["agent/type"] = torch.nn.functional.one_hot(ag_type, 3).bool()
    out["agent/cmd"] = torch.zeros(n_ag, 8, dtype=torch.bool)
    role = torch.zeros(n_ag, 3, dtype=torch.bool)
    role[0, 0] = True
    role[1 : min(4, n_ag), 2] = True
    out["agent/role"] = role
    out["agent/size"] = torch.tensor([4.5, 2.0, 1.6]).expand(n_ag, 3).clone()
    out["agent/goal"] = torch.cat([xy[:, -1], yaw[:, -1:], spd[:, None]], -1)
    dest = torch.zeros(n_ag, dtype=torch.int64)
    first_valid = mp_valid[:, 0]
    for a in range(n_ag):
        ok = torch.zeros(n_mp, dtype=torch.bool)
        for ty in _DEST_TYPES[int(ag_type[a])]:
            ok |= mp_type_idx == ty
        cand = torch.nonzero(ok & first_valid).flatten()
        dest[a] = cand[torch.randint(0, len(cand), (1,), generator=g)]
    out["agent/dest"] = dest
    return out


def to_history_batch(batch: Dict[str, torch.Tensor], n_step_hist: int = 11) -> Dict[str, torch.Tensor]:
    """Test-time view of an episode: `history/*` keys hold the first `n_step_hist` steps
    (reference data_h5_womd.py tensor_size_test); map keys are shared."""
    out = {k: v for k, v in batch.items() if k.startswith("map/")}
    for k, v in batch.items():
        if k.startswith(("agent/", "tl_lane/", "tl_stop/")):
            if v.dim() >= 3 and v.shape[2] == batch["agent/valid"].shape[2] and k.split("/")[1] in (
                "valid", "pos", "vel", "spd", "acc", "yaw_bbox", "yaw_rate", "state",
            ):
                v = v[:, :, :n_step_hist]
            out["history/" + k] = v
    return out


DATA_SIZE = {
    "agent/cmd": (64, 8), "agent/goal": (64, 4), "map/valid": (1024, 20), "map/type": (1024, 11),
    "tl_stop/state": (50, 91, 5), "agent/spd": (64, 91, 1), "agent/acc": (64, 91, 1), "agent/yaw_rate": (64, 91, 1),
    "agent/size": (64, 3), "agent/type": (64, 3),
}
