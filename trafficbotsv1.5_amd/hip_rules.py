"""ctypes wrappers of the rule-check / rollout-filter entry points (tbx_rule_*, tbx_filter_futures). Re-exported by hip.py."""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import load  # noqa: F401
from .hip_base import _check, _cptr, _ptr, stream_ptr


def rule_tables(mp_valid_u8, mp_type_idx_u8, mp_pos, mp_dir):
    """-> (seg [n,M*N,4], n_seg [n] i32, lane [n,M*N,2], n_lane [n] i32): compacted road-edge segments / lane-centre nodes."""
    n, M, N = mp_valid_u8.shape
    dev = mp_pos.device
    seg = torch.empty(n, M * N, 4, dtype=torch.float32, device=dev)
    lane = torch.empty(n, M * N, 2, dtype=torch.float32, device=dev)
    n_seg = torch.empty(n, dtype=torch.int32, device=dev)
    n_lane = torch.empty(n, dtype=torch.int32, device=dev)
    rc = load().tbx_rule_tables(_cptr(mp_valid_u8, torch.uint8), _cptr(mp_type_idx_u8, torch.uint8), _cptr(mp_pos, torch.float32),
                                _cptr(mp_dir, torch.float32), mp_pos.shape[-1], n, M, N, _ptr(seg), _ptr(n_seg), _ptr(lane),
                                _ptr(n_lane), stream_ptr())
    _check(rc, "tbx_rule_tables")
    return seg, n_seg, lane, n_lane


def rule_grid(seg, n_seg, lane, n_lane):
    """tbx_rule_grid: the tables of rule_tables sorted into a uniform raster -> dict(seg, lane: the sorted tables; seg_start, lane_start
    [n, cells + 1] i32; seg_grid, lane_grid [n, 4] f32) - what RuleCtx.seg / .lane / .seg_start / ... point at."""
    n, cap = seg.shape[:2]
    dev = seg.device
    cells = int(load().tbx_rule_grid_cells())
    out = dict(seg=torch.empty_like(seg), lane=torch.empty_like(lane), seg_start=torch.empty(n, cells + 1, dtype=torch.int32, device=dev),
               lane_start=torch.empty(n, cells + 1, dtype=torch.int32, device=dev), seg_grid=torch.empty(n, 4, dtype=torch.float32, device=dev),
               lane_grid=torch.empty(n, 4, dtype=torch.float32, device=dev))
    rc = load().tbx_rule_grid(_cptr(seg, torch.float32), _cptr(n_seg, torch.int32), _cptr(lane, torch.float32), _cptr(n_lane, torch.int32), n, cap,
                              _ptr(out["seg"]), _ptr(out["seg_start"]), _ptr(out["seg_grid"]), _ptr(out["lane"]), _ptr(out["lane_start"]),
                              _ptr(out["lane_grid"]), stream_ptr())
    _check(rc, "tbx_rule_grid")
    return out


def rule_check(ctx: RuleCtx, valid_u8, pose, motion, tl_state_u8, ld_t: int, t0: int, n_t: int, flags):
    rc = load().tbx_rule_check(C.byref(ctx), _cptr(valid_u8, torch.uint8), _cptr(pose, torch.float32), _cptr(motion, torch.float32),
                               _cptr(tl_state_u8, torch.uint8), ld_t, t0, n_t, _cptr(flags, torch.uint8), stream_ptr())
    _check(rc, "tbx_rule_check")


def rule_accumulate(raw, n_rows: int, ld_t: int, t0: int, n_t: int, acc_state, passive_counter, out_now, out_acc):
    rc = load().tbx_rule_accumulate(_cptr(raw, torch.uint8), n_rows, ld_t, t0, n_t, _cptr(acc_state, torch.uint8),
                                    _cptr(passive_counter, torch.float32), _cptr(out_now, torch.uint8), _cptr(out_acc, torch.uint8),
                                    stream_ptr())
    _check(rc, "tbx_rule_accumulate")


def rule_navi_check(valid_u8, pose, boundary, map_batch_div: int, dest: Optional[dict], goal, goal_thresh, acc, out_now):
    """tbx_rule_navi_check: outside-map / destination-reached / goal-reached of ONE step. dest = dict(invalid [n,A,N] u8, pos / dir
    [n,A,N,2] f32, kind [n,A] u8, thresh [n,A] f32) or None (no destinations); goal [n,A,4] + goal_thresh [n,A] or None; acc [3,n,A] u8
    (outside_map, dest_reached, goal_reached) updated in place; out_now [3,n,A] u8."""
    n, A = valid_u8.shape
    d = dest or {}
    rc = load().tbx_rule_navi_check(_cptr(valid_u8, torch.uint8), _cptr(pose, torch.float32), _cptr(boundary, torch.float32), map_batch_div,
                                    _cptr(d.get("invalid"), torch.uint8), _cptr(d.get("pos"), torch.float32), _cptr(d.get("dir"), torch.float32),
                                    _cptr(d.get("kind"), torch.uint8), _cptr(d.get("thresh"), torch.float32), _cptr(goal, torch.float32),
                                    _cptr(goal_thresh, torch.float32), n, A, d["invalid"].shape[2] if dest else 0, _cptr(acc, torch.uint8),
                                    _cptr(out_now, torch.uint8), stream_ptr())
    _check(rc, "tbx_rule_navi_check")


def filter_futures(flags, col_bit: int, ag_role_any, n_scene: int, n_k: int, t_start: int, w_road_edge: float, n_keep: int,
                   pred_pose=None):
    """flags [n_scene*n_k, A, T] u8 bits, ag_role_any [n_scene, A] u8 -> (score [n_scene,n_k], idx [n_scene,n_keep] i32,
    trajs [n_scene, n_keep, A, T - t_start, 3] or None)."""
    A, T = flags.shape[-2:]
    dev = flags.device
    score = torch.empty(n_scene, n_k, dtype=torch.float32, device=dev)
    idx = torch.empty(n_scene, n_keep, dtype=torch.int32, device=dev)
    trajs = None if pred_pose is None else torch.empty(n_scene, n_keep, A, T - t_start, 3, dtype=torch.float32, device=dev)
    rc = load().tbx_filter_futures(_cptr(flags, torch.uint8), col_bit, _cptr(ag_role_any, torch.uint8), n_scene, n_k, A, T, t_start,
                                   w_road_edge, n_keep, _ptr(score), _ptr(idx), _cptr(pred_pose, torch.float32), _ptr(trajs),
                                   stream_ptr())
    _check(rc, "tbx_filter_futures")
    return score, idx, trajs
