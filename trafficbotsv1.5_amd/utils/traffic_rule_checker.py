"""`TrafficRuleChecker` (utils/traffic_rule_checker.py:10-451) on the HIP rule kernels.

The reference calls `check` once per Python step of the rollout (pl_modules/waymo_motion.py:250). Two of its checks feed
back into the simulation (outside-map -> disable agent, destination reached -> disable navi) and run inside
`tbx_sim_step`; the other five (collided, collided_wosac, run_road_edge, run_red_light, passive) only produce metrics and
the WOSAC rollout filter, so here they are evaluated for ALL steps of a finished rollout at once from the device-resident
rollout log (`check_log`, one `tbx_rule_check` launch over n_rollout x n_step frames + one `tbx_rule_accumulate`), which
fills a 256-CU device where 80 per-step calls on 64-128 agents would not. `check` keeps the reference's per-step
signature and its complete result (a one-frame log for the five + `tbx_rule_navi_check` for outside_map / dest_reached, which a
step-wise caller hands to `Dynamics.disable_ag / disable_navi` as the reference's `rollout` does, waymo_motion.py:250,304-305).
Constructor arguments are the reference's; map tensors may be given per scene while the agent tensors are per rollout
(n_rollout = K * n_scene) - or per rollout too, as the reference's `joint_future_pred` repeats them (:497-503).

There is no CPU path: every method needs device tensors and the HIP library.
"""
from typing import Dict, Optional

import torch
from torch import Tensor

from .. import hip

def dest_tables(ag_dest: Tensor, mp_valid: Tensor, mp_type: Tensor, mp_pos: Tensor, mp_dir: Tensor) -> Dict[str, Tensor]:
    """TrafficRuleChecker._get_dest (traffic_rule_checker.py:87-107) in the form tbx_sim_step / tbx_rule_navi_check read:
    ag_dest [n, A] int64 polyline index; map tensors [n / div, M, N, ..] (div rollouts share a scene's polylines) ->
    invalid [n,A,N] u8, pos / dir [n,A,N,2] f32 (dir normalised), kind [n,A] u8 (1 lane, 2 road edge), thresh [n,A] f32 (50 m / 10 m)."""
    n = ag_dest.shape[0]
    u8 = torch.uint8
    bsel = (torch.arange(n, device=ag_dest.device) // (n // mp_valid.shape[0])).unsqueeze(1)
    d_type = mp_type[bsel, ag_dest]                      # [n, A, 11]
    d_dir = mp_dir[bsel, ag_dest][..., :2].float()
    d_dir = d_dir / torch.norm(d_dir, dim=-1, keepdim=True)
    return {"pos": mp_pos[bsel, ag_dest][..., :2].float().contiguous(), "dir": d_dir.contiguous(),
            "invalid": (~mp_valid[bsel, ag_dest]).to(u8).contiguous(),
            "kind": (d_type[:, :, :4].any(-1).to(u8) + 2 * d_type[:, :, 4].to(u8)).contiguous(),
            "thresh": (50.0 * (1 - d_type[:, :, 4].float() * 0.8)).contiguous()}


_KEYS = (("collided", hip.RULE_COLLIDED), ("collided_wosac", hip.RULE_COLLIDED_WOSAC), ("run_road_edge", hip.RULE_RUN_ROAD_EDGE),
         ("run_red_light", hip.RULE_RUN_RED_LIGHT), ("passive", hip.RULE_PASSIVE))


class TrafficRuleChecker:
    def __init__(self, mp_boundary: Tensor, mp_valid: Tensor, mp_type: Tensor, mp_pos: Tensor, mp_dir: Tensor, ag_type: Tensor,
                 ag_size: Tensor, ag_goal: Optional[Tensor], ag_dest: Optional[Tensor], tl_valid: Tensor, tl_pose: Tensor,
                 disable_check: bool, collision_size_scale: float = 1.1) -> None:
        self.mp_boundary, self.mp_valid, self.mp_type = mp_boundary, mp_valid, mp_type
        self.mp_pos, self.mp_dir = mp_pos, mp_dir
        self.ag_type, self.ag_size, self.ag_goal, self.ag_dest = ag_type, ag_size, ag_goal, ag_dest
        self.tl_valid, self.tl_pose, self.disable_check = tl_valid, tl_pose, disable_check
        self.collision_size_scale = collision_size_scale
        # the scene tables sorted into a uniform raster (tbx_rule_grid): a vehicle's road-edge / lane tests visit the few cells around it
        # instead of the scene's ~6,400 rows - bit-identical flags (tests/test_hip_rules.py runs both); False: the full scans
        self.use_grid = True
        self._bits: Optional[Tensor] = None
        self._ctx: Optional[hip.RuleCtx] = None
        self._keep = None
        self._acc: Optional[Tensor] = None          # running OR of the five flags [n, A] u8 bits
        self._navi: Optional[dict] = None           # check(): the destination tables + the outside_map / dest_reached accumulators
        self.passive_counter: Optional[Tensor] = None  # [n, A] f32 (traffic_rule_checker.py:42)

    # ------------------------------------------------------------------ static tables (once per scene batch)
    @torch.no_grad()
    def _setup(self) -> hip.RuleCtx:
        if self._ctx is not None:
            return self._ctx
        n, A = self.ag_type.shape[:2]
        n_scene = self.mp_valid.shape[0]
        assert n % n_scene == 0, "agent tensors must hold a whole number of rollouts per scene"
        u8 = torch.uint8
        seg, n_seg, lane, n_lane = hip.rule_tables(self.mp_valid.to(u8).contiguous(), self.mp_type.to(u8).argmax(-1).to(u8).contiguous(),
                                                   self.mp_pos.float().contiguous(), self.mp_dir.float().contiguous())
        size = self.ag_size.float().contiguous()
        if size.shape[-1] != 3:
            size = torch.cat([size, size.new_zeros(*size.shape[:-1], 3 - size.shape[-1])], -1).contiguous()
        grid = hip.rule_grid(seg, n_seg, lane, n_lane) if self.use_grid else {}
        seg, lane = grid.get("seg", seg), grid.get("lane", lane)  # (the sorted tables: every consumer is an any())
        keep = dict(seg=seg, n_seg=n_seg, lane=lane, n_lane=n_lane, ag_size=size, **{k: v for k, v in grid.items() if k not in ("seg", "lane")},
                    ag_type_idx=self.ag_type.to(u8).argmax(-1).to(u8).contiguous(), tl_valid=self.tl_valid.to(u8).contiguous(),
                    tl_pose=self.tl_pose.float().contiguous())
        ctx = hip.RuleCtx()
        ctx.n_batch, ctx.n_ag, ctx.n_tl = n, A, self.tl_valid.shape[1]
        ctx.map_batch_div, ctx.cap = n // n_scene, seg.shape[1]
        for k, v in keep.items():
            setattr(ctx, k, v.data_ptr())
        ctx.collision_size_scale = self.collision_size_scale
        # the five flag bits as a device tensor made ON the device (a host list -> device copy would wait for the stream to drain)
        assert [b for _, b in _KEYS] == [1 << i for i in range(len(_KEYS))]
        self._bits = (1 << torch.arange(len(_KEYS), device=size.device, dtype=torch.int32)).to(u8).view(-1, 1, 1, 1, 1)
        self._keep, self._ctx = keep, ctx
        self._acc = torch.zeros(n, A, dtype=u8, device=size.device)
        self.passive_counter = torch.zeros(n, A, dtype=torch.float32, device=size.device)
        return ctx

    # ------------------------------------------------------------------ whole rollout log
    @torch.no_grad()
    def check_log(self, valid: Tensor, pose: Tensor, motion: Tensor, tl_state_bits: Tensor, t0: int = 0,
                  n_t: Optional[int] = None) -> Dict[str, Tensor]:
        """valid [n,A,T] u8/bool, pose / motion [n,A,T,3], tl_state_bits [n,L,T] u8 (5-bit state masks): steps [t0, t0+n_t)
        continue from the accumulated state of earlier calls. -> the reference's violation dict restricted to the five
        metric-only checks, every entry [n, A, T] bool (`k` accumulated, `k_this_step` per step)."""
        ctx = self._setup()
        n, A, T = valid.shape
        n_t = T - t0 if n_t is None else n_t
        if self.disable_check:  # training: the reference returns its (all-false) accumulators (:357-404)
            z = torch.zeros(n, A, T, dtype=torch.bool, device=pose.device)
            return {k + s: z for k, _ in _KEYS for s in ("", "_this_step")}
        v8 = valid.to(torch.uint8).contiguous()
        # (the kernels write every frame of [t0, t0 + n_t): a whole-log call needs no zero fill)
        alloc = torch.empty if (t0 == 0 and n_t == T) else torch.zeros
        both = alloc(2, n, A, T, dtype=torch.uint8, device=pose.device)
        now, acc = both[0], both[1]
        hip.rule_check(ctx, v8, pose.float().contiguous(), motion.float().contiguous(), tl_state_bits.contiguous(), T, t0, n_t, now)
        hip.rule_accumulate(now, n * A, T, t0, n_t, self._acc, self.passive_counter, now, acc)
        # the ten boolean fields as views of ONE [5 flags, 2 (this step | accumulated), n, A, T] tensor: two launches instead of twenty
        flags = (both.unsqueeze(0) & self._bits) != 0
        out = {}
        for i, (k, _) in enumerate(_KEYS):
            out[k] = flags[i, 1]
            out[k + "_this_step"] = flags[i, 0]
        return out

    # ------------------------------------------------------------------ the reference's per-step entry point
    def _navi_setup(self) -> dict:
        if self._navi is None:
            n, A = self.ag_type.shape[:2]
            dev = self.ag_type.device
            dest = None if self.ag_dest is None else dest_tables(self.ag_dest, self.mp_valid, self.mp_type, self.mp_pos, self.mp_dir)
            goal = None if self.ag_goal is None else self.ag_goal.float().contiguous()
            self._navi = dict(dest=dest, boundary=self.mp_boundary.float().contiguous(), div=n // self.mp_boundary.shape[0], goal=goal,
                              goal_thresh=None if goal is None else (self.ag_size[:, :, 0].float() * 8).contiguous(),  # (:66, un-scaled length)
                              acc=torch.zeros(3, n, A, dtype=torch.uint8, device=dev))
        return self._navi

    @torch.no_grad()
    def check(self, valid: Tensor, pose: Tensor, motion: Tensor, tl_state: Tensor) -> Dict[str, Tensor]:
        """traffic_rule_checker.py:342-451 for one step: valid [n,A] bool, pose / motion [n,A,3], tl_state [n,L,5] one-hot
        bool -> the reference's sixteen entries {outside_map, outside_map_this_step, collided, ..., goal_reached, dest_reached,
        dest_reached_this_step}, [n,A] bool, accumulating across calls like the reference's object."""
        w = 1 << torch.arange(tl_state.shape[-1], device=tl_state.device, dtype=torch.int32)
        bits = (tl_state.to(torch.int32) * w).sum(-1).to(torch.uint8)
        out = self.check_log(valid.unsqueeze(-1), pose.unsqueeze(2), motion.unsqueeze(2), bits.unsqueeze(-1))
        out = {k: v[..., 0] for k, v in out.items()}
        nv = self._navi_setup()
        now = torch.empty_like(nv["acc"])
        hip.rule_navi_check(valid.to(torch.uint8).contiguous(), pose.float().contiguous(), nv["boundary"], nv["div"], nv["dest"], nv["goal"],
                            nv["goal_thresh"], nv["acc"], now)
        acc = nv["acc"].clone().view(torch.bool)  # (the reference hands out its accumulators as of THIS step: `self.x = self.x | now` rebinds)
        now = now.view(torch.bool)
        out.update(outside_map=acc[0], outside_map_this_step=now[0], dest_reached=acc[1], dest_reached_this_step=now[1],
                   goal_reached=acc[2], goal_reached_this_step=now[2])
        return out
