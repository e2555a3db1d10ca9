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


def make_edge_batch(n_ag: int = 8, n_mp: int = 64, n_tl: int = 8, seed: int = 0, kind: str = "mixed") -> Dict[str, torch.Tensor]:
    """A batch of three scenes with the domain's empty inputs (the training-step fixture `train_c1_edge.npz` and its test are made from
    it): scene 0 without a valid traffic light, scene 1 with TWO agents, scene 2 without a valid polyline. Everything else as
    make_scene. kind = "no_lights": ONE scene without a valid light - the light-state term's counter is zero and the reference leaves
    the term out of the loss (`train_c1_nolights.npz`)."""
    if kind == "no_lights":
        b = make_scene(1, n_ag, n_mp, n_tl, seed=seed)
        b["tl_lane/valid"][:] = False
        b["tl_stop/valid"][:] = False
        return b
    b = make_scene(3, n_ag, n_mp, n_tl, seed=seed)
    b["tl_lane/valid"][0] = False
    b["tl_stop/valid"][0] = False
    b["agent/valid"][1, 2:] = False
    b["map/valid"][2] = False
    return b


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


def make_rule_episode(n_sc: int = 2, n_ag: int = 16, n_mp: int = 64, n_tl: int = 8, n_step: int = 40, seed: int = 0,
                      n_pl_node: int = 20, extent: float = 60.0) -> Dict[str, torch.Tensor]:
    """A crowded little world for the traffic-rule checks: agents start on map polylines, drive along
    them in platoons (so boxes overlap, road edges get crossed, stop points get run over, slow vehicles idle on lanes),
    with ragged validity and all three agent types. Returns the TrafficRuleChecker constructor tensors plus a
    [n_sc, n_ag, n_step] episode (valid, pose, motion) and per-step light states."""
    scenes = []
    for i in range(n_sc):
        g = torch.Generator().manual_seed(10_000 + seed + i)
        U = lambda *s: torch.rand(*s, generator=g)
        N = lambda *s: torch.randn(*s, generator=g)
        start = (U(n_mp, 2) - 0.5) * 2 * extent
        head = ((U(n_mp) - 0.5) * 2 * math.pi)[:, None] + torch.cat([torch.zeros(n_mp, 1), N(n_mp, n_pl_node - 1) * 0.03], 1).cumsum(1)
        seg = torch.stack([head.cos(), head.sin()], -1)
        pos = start[:, None] + torch.cat([torch.zeros(n_mp, 1, 2), seg[:, :-1].cumsum(1)], 1)
        mp_valid = torch.arange(n_pl_node)[None] < torch.randint(1, n_pl_node + 1, (n_mp, 1), generator=g)
        ty = torch.randint(0, 11, (n_mp,), generator=g)
        ty[:8] = torch.tensor([0, 1, 2, 4, 5, 7, 0, 4])
        mp_valid[:8] = True
        tl_idx = torch.cat([torch.tensor([0, 1, 6]), 8 + torch.randperm(n_mp - 8, generator=g)[: n_tl - 3]])[:n_tl]
        tl_state = torch.nn.functional.one_hot(
            torch.randint(0, 5, (n_tl, (n_step + 4) // 5), generator=g).repeat_interleave(5, 1)[:, :n_step], 5).bool()
        tl_state[0, :, :] = torch.tensor([False, True, False, False, False])  # light 0 stays on STOP
        # agents: platoons of 2 on a polyline, the follower faster than the leader
        lane = torch.randint(0, n_mp, (n_ag,), generator=g)
        lane[: min(6, n_ag)] = torch.tensor([0, 0, 1, 1, 6, 6])[: min(6, n_ag)]
        lane[1::2] = lane[0::2][: len(lane[1::2])]
        node = torch.randint(0, n_pl_node // 2, (n_ag,), generator=g)
        node[1::2] = node[0::2][: len(node[1::2])] + 4
        node[: min(6, n_ag)] = torch.tensor([0, 4, 0, 5, 0, 3])[: min(6, n_ag)]  # start at / just past the lights' stop points
        xy0 = pos[lane, node] + N(n_ag, 2) * 0.6
        back = torch.zeros(n_ag)
        back[: min(6, n_ag) : 2] = 6.0  # the lights' followers start 6 m before the stop point
        xy0 = xy0 - back[:, None] * seg[lane, node]
        yaw0 = head[lane, node] + N(n_ag) * 0.15
        spd = U(n_ag) * 6.0
        spd[0::2] += 6.0
        yaw_rate = N(n_ag) * 0.05
        t = torch.arange(n_step, dtype=torch.float32) * 0.1
        yaw = yaw0[:, None] + yaw_rate[:, None] * t[None]
        vel = spd[:, None, None] * torch.stack([yaw.cos(), yaw.sin()], -1)
        xy = xy0[:, None] + torch.cat([torch.zeros(n_ag, 1, 2), (vel[:, :-1] * 0.1).cumsum(1)], 1)
        valid = U(n_ag, n_step) > 0.05
        valid[n_ag - 1] = False
        valid[n_ag - 2, n_step // 2:] = False
        at = torch.randint(0, 3, (n_ag,), generator=g)
        at[: min(8, n_ag)] = 0
        base = torch.tensor([[4.5, 2.0, 1.6], [0.8, 0.8, 1.7], [1.8, 0.6, 1.6]])[at]
        scenes.append({
            "map/valid": mp_valid, "map/type": torch.nn.functional.one_hot(ty, 11).bool(),
            "map/pos": torch.cat([pos, torch.zeros(n_mp, n_pl_node, 1)], -1), "map/dir": torch.cat([seg, torch.zeros(n_mp, n_pl_node, 1)], -1),
            "map/boundary": torch.tensor([-200.0, 200.0, -200.0, 200.0]),
            "tl/valid": U(n_tl) < 0.8, "tl/pose": torch.cat([pos[tl_idx, 0], head[tl_idx, :1]], -1), "tl/state": tl_state,
            "agent/type": torch.nn.functional.one_hot(at, 3).bool(), "agent/size": base * (0.8 + 0.4 * U(n_ag, 1)),
            "agent/valid": valid, "agent/pose": torch.cat([xy, yaw[..., None]], -1),
            "agent/motion": torch.stack([spd[:, None].expand(-1, n_step), torch.zeros(n_ag, n_step), yaw_rate[:, None].expand(-1, n_step)], -1),
        })
        scenes[-1]["tl/valid"][0] = True
    return {k: torch.stack([s[k] for s in scenes], 0) for k in scenes[0]}


def make_filter_case(n_sc: int = 2, n_k: int = 48, n_ag: int = 12, n_step: int = 30, seed: int = 0, p_col: float = 0.02,
                     p_edge: float = 0.03) -> Dict[str, torch.Tensor]:
    """Seeded inputs of the WOSAC rollout filter: K joint futures per scene with sparse, accumulated
    (monotone in time) collision / road-edge flags and random trajectories."""
    g = torch.Generator().manual_seed(20_000 + seed)
    first = lambda p: (torch.rand(n_sc, n_k, n_ag, n_step, generator=g) < p / n_step * 4).cummax(-1)[0]
    return {"pred_pose": torch.randn(n_sc, n_k, n_ag, n_step, 3, generator=g) * 30.0, "collided": first(p_col), "collided_wosac": first(p_col),
            "run_road_edge": first(p_edge), "ag_role": torch.rand(n_sc, n_ag, 3, generator=g) < 0.3}
