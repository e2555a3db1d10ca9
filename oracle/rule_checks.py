"""ORACLE (test infrastructure, never on the product path): CPU restatement of the reference's per-step traffic-rule
checks, SURVEY.md §8f row 1.

Follows `utils/traffic_rule_checker.py` (TrafficRuleChecker.__init__ :11-85, _check_collided :122-156,
_check_run_road_edge :158-181, _check_run_red_light :183-233, _check_passive :235-298, check :342-451,
_get_road_edge :453-484, _get_lane_center :486-501, ccw :504-505) and `utils/wosac_collision.py` (get_ag_bbox :20-47,
_signed_distance_from_point_to_convex_polygon :50-116, _get_edge_info :119-141, _get_downmost_edge_in_box :144-179,
_minkowski_sum_of_box_and_box_points :182-208, check_collided_wosac :211-257) of the reference, as pure functions of
explicit state. fp32 torch CPU ops in the reference's operation order, so every comparison sees the same roundings.

Pinned: tests/test_oracle_rules.py checks it against tests/golden/rules.npz, which holds the outputs of the reference's
own TrafficRuleChecker on seeded scenes (tests/golden/make_golden.py: gen_rules).
"""
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

BIG = 1e10            # wosac_collision.py:9 EXTREMELY_LARGE_DISTANCE
CORNER_ROUNDING = 0.7  # wosac_collision.py:17


# ---------------------------------------------------------------------------------------------- geometry
def ag_bbox(pose: Tensor, size_lw: Tensor) -> Tensor:
    """wosac_collision.py:20-47. pose [n,A,3], size_lw [n,A,2] -> corners [n,A,4,2] (front-left, rear-left, rear-right,
    front-right: counter-clockwise)."""
    c, s = torch.cos(pose[..., 2]), torch.sin(pose[..., 2])
    f = torch.stack([c, s], -1)
    r = torch.stack([s, -c], -1)
    of = 0.5 * size_lw[..., [0]].expand(-1, -1, 2) * f
    orr = 0.5 * size_lw[..., [1]].expand(-1, -1, 2) * r
    off = torch.stack([of - orr, -of - orr, -of + orr, of + orr], 2)
    return pose[:, :, None, :2].expand(-1, -1, 4, -1) + off


def _ccw(A: Tensor, B: Tensor, C: Tensor) -> Tensor:
    """traffic_rule_checker.py:504-505."""
    return (C[..., 1] - A[..., 1]) * (B[..., 0] - A[..., 0]) > (B[..., 1] - A[..., 1]) * (C[..., 0] - A[..., 0])


# ---------------------------------------------------------------------------------------------- static tables
def road_edges(mp_valid: Tensor, mp_type: Tensor, mp_pos: Tensor, mp_dir: Tensor) -> Tuple[Tensor, Tensor]:
    """traffic_rule_checker.py:453-484: node segments (pos, pos + dir) of polylines of type 4 / 5 / 7."""
    ok = mp_valid & mp_type[:, :, [4, 5, 7]].any(-1, keepdim=True)
    seg = torch.stack([mp_pos[..., :2], mp_pos[..., :2] + mp_dir[..., :2]], -2)
    return seg.flatten(1, 2), ok.flatten(1, 2)


def lane_centers(mp_valid: Tensor, mp_type: Tensor, mp_pos: Tensor) -> Tuple[Tensor, Tensor]:
    """traffic_rule_checker.py:486-501: nodes of polylines of type 0 / 1 / 2."""
    ok = mp_valid & mp_type[:, :, :3].any(-1, keepdim=True)
    return mp_pos[..., :2].flatten(1, 2), ok.flatten(1, 2)


# ---------------------------------------------------------------------------------------------- checks
def check_collided(valid: Tensor, bbox: Tensor, ag_type: Tensor) -> Tensor:
    """traffic_rule_checker.py:122-156 with the mask of :45-49 (no self pairs, no ped-ped pairs)."""
    n, A = valid.shape
    nxt = bbox.roll(-1, 2)
    line = torch.cat([nxt[..., [1]] - bbox[..., [1]], bbox[..., [0]] - nxt[..., [0]],
                      nxt[..., [0]] * bbox[..., [1]] - nxt[..., [1]] * bbox[..., [0]]], -1)       # [n,A,4,3]
    pt = torch.cat([bbox, torch.ones_like(bbox[..., [0]])], -1)                                   # [n,A,4,3]
    line = line[:, :, None, :, None, :].expand(-1, -1, A, -1, 4, -1)
    pt = pt[:, None, :, None, :, :].expand(-1, A, -1, 4, -1, -1)
    outside = torch.sum(line * pt, -1) > 0                                                        # [n,A,A,4,4]
    free = outside.all(-1).any(-1)
    free = free | free.transpose(1, 2)
    ped = ag_type[:, :, 1]
    skip = torch.eye(A, dtype=torch.bool)[None] | (ped.unsqueeze(1) & ped.unsqueeze(2)) | ~(valid[:, :, None] & valid[:, None, :])
    return ~((free | skip).all(-1))


def _edge_info(poly: Tensor):
    """wosac_collision.py:119-141."""
    e = poly.roll(-1, 2) - poly
    ln = torch.norm(e, dim=-1)
    t = e / ln.unsqueeze(-1)
    return t, torch.stack([-t[..., 1], t[..., 0]], -1), ln


def _signed_distance_origin(poly: Tensor) -> Tensor:
    """wosac_collision.py:50-116 with the query point at the origin. poly [n,P,8,2] -> [n,P]."""
    t, nrm, ln = _edge_info(poly)
    q = torch.zeros_like(poly[:, :, 0, :]).unsqueeze(2) - poly
    d_vert = torch.norm(q, dim=-1)
    perp = torch.sum(-nrm * q, -1)
    inside = torch.all(perp <= 0, -1)
    prop = torch.sum(t * q, -1) / ln
    on_edge = (prop >= 0.0) & (prop <= 1.0)
    ap = perp.abs()
    d_edge = torch.where(on_edge, ap, torch.zeros_like(ap) + BIG)
    m = torch.amin(torch.cat([d_edge, d_vert], -1), -1)
    return torch.where(inside, -m, m)


def _downmost_edge(box: Tensor):
    """wosac_collision.py:144-179."""
    i0 = torch.argmin(box[..., 1], -1).unsqueeze(-1)
    bi = torch.arange(box.shape[0])[:, None, None]
    pi = torch.arange(box.shape[1])[None, :, None]
    e = box[bi, pi, torch.remainder(i0 + 1, 4)] - box[bi, pi, i0]
    return i0, e / torch.norm(e, dim=-1).unsqueeze(-1)


def _minkowski(b1: Tensor, b2: Tensor) -> Tensor:
    """wosac_collision.py:182-208: the 8-vertex Minkowski sum of two boxes (each [n,P,4,2], counter-clockwise)."""
    bi = torch.arange(b1.shape[0])[:, None, None]
    pi = torch.arange(b1.shape[1])[None, :, None]
    o1 = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3])
    o2 = torch.tensor([0, 1, 1, 2, 2, 3, 3, 0])
    s1, d1 = _downmost_edge(b1)
    s2, d2 = _downmost_edge(b2)
    cond = ((d1[..., 0] * d2[..., 1] - d1[..., 1] * d2[..., 0]) >= 0.0).expand(-1, -1, 8)
    p1 = b1[bi, pi, torch.remainder(torch.where(cond, o2, o1) + s1, 4)]
    p2 = b2[bi, pi, torch.remainder(torch.where(cond, o1, o2) + s2, 4)]
    return p1 + p2


def wosac_signed_distance(pose: Tensor, size: Tensor) -> Tensor:
    """wosac_collision.py:228-250: pairwise signed distance of the rounded boxes, [n,A,A] (no masks applied)."""
    n, A, _ = pose.shape
    shrink = torch.min(size[:, :, 0], size[:, :, 1]) * CORNER_ROUNDING / 2.0
    box = ag_bbox(pose, size[:, :, :2] - 2.0 * shrink.unsqueeze(-1))
    ev = box.unsqueeze(2).expand(-1, -1, A, -1, -1).flatten(1, 2)
    al = box.unsqueeze(1).expand(-1, A, -1, -1, -1).flatten(1, 2)
    sd = _signed_distance_origin(_minkowski(ev, -1.0 * al)).view(n, A, A)
    sd = sd - shrink.unsqueeze(1)
    return sd - shrink.unsqueeze(2)


def check_collided_wosac(pose: Tensor, size: Tensor, valid: Tensor) -> Tensor:
    """wosac_collision.py:211-257."""
    A = pose.shape[1]
    sd = wosac_signed_distance(pose, size)
    bad = ~(valid.unsqueeze(1) & valid.unsqueeze(2)) | torch.eye(A, dtype=torch.bool)[None]
    return torch.amin(sd.masked_fill(bad, BIG), 2) < 0.0


def check_run_road_edge(valid: Tensor, bbox: Tensor, veh: Tensor, seg: Tensor, seg_ok: Tensor) -> Tensor:
    """traffic_rule_checker.py:158-181: any box edge properly crossing any road-edge node segment (vehicles only)."""
    a = bbox.unsqueeze(2)                # [n,A,1,4,2]
    b = bbox.roll(-1, 2).unsqueeze(2)
    c, d = seg[:, None, :, None, 0], seg[:, None, :, None, 1]   # [n,1,S,1,2]
    hit = (_ccw(a, c, d) != _ccw(b, c, d)) & (_ccw(a, b, c) != _ccw(a, b, d))
    return (hit.any(-1) & seg_ok.unsqueeze(1)).any(-1) & valid & veh


def check_run_red_light(valid, pose, motion, tl_valid, tl_pose, tl_state, half_len, half_wid, veh) -> Tensor:
    """traffic_rule_checker.py:183-233: a STOP light's stop point is inside the agent's (shrunk / widened) footprint now
    and outside after 0.1 s at the current speed. half_len / half_wid [n,A,1]."""
    c, s = torch.cos(pose[..., 2]), torch.sin(pose[..., 2])
    f = torch.stack([c, s], -1).unsqueeze(2)
    r = torch.stack([s, -c], -1).unsqueeze(2)
    p0 = pose[..., :2].unsqueeze(2)
    p1 = p0 + 0.1 * motion[..., [0]].unsqueeze(2) * f
    t = tl_pose[:, None, :, :2]
    ins = lambda p: (torch.abs(torch.sum((t - p) * f, -1)) < half_len) & (torch.abs(torch.sum((t - p) * r, -1)) < half_wid)
    hit = ins(p0) & ~ins(p1) & (valid & veh).unsqueeze(2) & (tl_valid & tl_state[:, :, 1]).unsqueeze(1)
    return hit.any(-1)


def check_passive_raw(valid, pose, motion, tl_valid, tl_pose, tl_state, lane, lane_ok, veh) -> Tensor:
    """traffic_rule_checker.py:235-291 up to (not including) the counter: a slow vehicle on a lane with nothing ahead."""
    A = valid.shape[1]
    near_lane = ((torch.norm(pose[:, :, :2].unsqueeze(2) - lane.unsqueeze(1), dim=-1) < 2) & lane_ok.unsqueeze(1)).any(-1)
    slow = motion[:, :, 0] < 5
    f = torch.stack([torch.cos(pose[..., 2]), torch.sin(pose[..., 2])], -1).unsqueeze(2)
    tl_on = (tl_valid & tl_state[:, :, [0, 1, 2, 4]].any(-1)).unsqueeze(1)
    v = tl_pose[:, None, :, :2] - pose[:, :, :2].unsqueeze(2)
    vn = torch.norm(v, dim=-1)
    red_ahead = ((vn < 10) & (((f * v).sum(-1) / vn) > 0.95) & tl_on).any(-1)
    w = pose[:, :, :2].unsqueeze(1) - pose[:, :, :2].unsqueeze(2)
    wn = torch.norm(w, dim=-1)
    ag_ahead = ((wn < 10) & (((f * w).sum(-1) / wn) > 0.95) & valid.unsqueeze(1) & valid.unsqueeze(2)
                & ~torch.eye(A, dtype=torch.bool)[None]).any(-1)
    return valid & veh & near_lane & slow & ~red_ahead & ~ag_ahead


# ---------------------------------------------------------------------------------------------- the per-step checker
def check_goal_reached(valid: Tensor, pose: Tensor, goal: Tensor, goal_reached: Tensor, ag_length: Tensor) -> Tensor:
    """traffic_rule_checker.py:277-288 (_check_goal_reached; goal_thresh_pos = 8 agent lengths :66, goal_thresh_rot = 15 deg :67;
    cast_rad = transform_utils.py:9-11) -> goal_reached_this_step [n_sc, n_ag] bool."""
    import math

    pos_reached = torch.norm(pose[..., :2] - goal[..., :2], dim=-1) < ag_length * 8
    d = pose[..., 2] - goal[..., 2]
    rot_reached = torch.abs((d + math.pi) % (2 * math.pi) - math.pi) < math.radians(15)
    return pos_reached & rot_reached & valid & (~goal_reached)


class RuleCheckOracle:
    """The metric-only half of TrafficRuleChecker.check (traffic_rule_checker.py:342-451): collided, collided_wosac,
    run_road_edge, run_red_light, passive - per step and accumulated. (outside_map / dest_reached feed back into the
    rollout and are part of trafficbots_oracle.Sim.)"""

    KEYS = ("collided", "collided_wosac", "run_road_edge", "run_red_light", "passive")

    def __init__(self, mp_valid, mp_type, mp_pos, mp_dir, ag_type, ag_size, tl_valid, tl_pose, collision_size_scale: float = 1.1):
        self.size3 = ag_size
        self.size_scaled = ag_size[..., :2] * collision_size_scale          # :27
        self.ag_type, self.veh = ag_type, ag_type[:, :, 0]
        self.seg, self.seg_ok = road_edges(mp_valid, mp_type, mp_pos, mp_dir)
        self.lane, self.lane_ok = lane_centers(mp_valid, mp_type, mp_pos)
        self.half_len = ag_size[:, :, [0]] * 0.5 * 0.6                       # :55
        self.half_wid = ag_size[:, :, [1]] * 0.5 * 1.8                       # :56
        self.tl_valid, self.tl_pose = tl_valid, tl_pose
        z = torch.zeros_like(self.veh)
        self.acc = {k: z.clone() for k in self.KEYS}
        self.passive_counter = torch.zeros_like(self.veh, dtype=torch.float32)

    @torch.no_grad()
    def check(self, valid: Tensor, pose: Tensor, motion: Tensor, tl_state: Tensor) -> Dict[str, Tensor]:
        bbox = ag_bbox(pose, self.size_scaled)
        now = {
            "collided": check_collided(valid, bbox, self.ag_type),
            # NB the reference passes the SCALED size here (traffic_rule_checker.py:362 uses self.ag_size)
            "collided_wosac": check_collided_wosac(pose, self.size_scaled, valid),
            "run_road_edge": check_run_road_edge(valid, bbox, self.veh, self.seg, self.seg_ok),
            "run_red_light": check_run_red_light(valid, pose, motion, self.tl_valid, self.tl_pose, tl_state, self.half_len,
                                                 self.half_wid, self.veh),
        }
        raw = check_passive_raw(valid, pose, motion, self.tl_valid, self.tl_pose, tl_state, self.lane, self.lane_ok, self.veh)
        self.passive_counter = (self.passive_counter + raw) * raw             # :294
        now["passive"] = self.passive_counter > 20
        out = {}
        for k in self.KEYS:
            self.acc[k] = self.acc[k] | now[k]
            out[k], out[k + "_this_step"] = self.acc[k], now[k]
        return out
