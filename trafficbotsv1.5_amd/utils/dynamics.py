"""`Dynamics` / `MultiPathPP` (utils/dynamics.py:13-274). The state update itself (tanh-bounded action -> MultiPath++ midpoint
integration -> overrides -> disabling) is `tbx_sim_step`; the state lives in the rollout engine's device buffers. For step-wise
drivers (`WaymoMotion.forward`) this object exposes that state under the reference's attribute names and the two methods the
reference's `rollout` calls between two `forward`s (disable_ag / disable_navi)."""
from typing import Dict, Optional, Tuple

from torch import Tensor


class MultiPathPP:
    def __init__(self, dt: float = 0.1, max_acc: float = 4, max_yaw_rate: float = 1, **_) -> None:
        self.dt, self._max_acc, self._max_yaw_rate = dt, float(max_acc), float(max_yaw_rate)


class Dynamics:
    def __init__(self, veh, ped, cyc, navi_mode: str, use_veh_dynamics_for_all: bool = False) -> None:
        if use_veh_dynamics_for_all:
            raise NotImplementedError("per-type dynamics is the default")
        self.dt, self.action_dim, self.navi_mode = 0.1, 2, navi_mode
        strip = lambda c: {k: v for k, v in dict(c).items() if k != "_target_"}
        # tuple order = agent type one-hot order (veh, ped, cyc): dynamics.py:23-27
        self.ag_dynamics: Tuple[MultiPathPP, ...] = tuple(MultiPathPP(dt=self.dt, **strip(c)) for c in (veh, ped, cyc))

    @property
    def max_acc(self):
        return [d._max_acc for d in self.ag_dynamics]

    @property
    def max_yaw_rate(self):
        return [d._max_yaw_rate for d in self.ag_dynamics]

    # ------------------------------------------------------------------ the reference's step-0 call (dynamics.py:29-64)
    _eng = None
    _pending: Optional[dict] = None   # init()'s arguments until the first WaymoMotion.forward builds the device state from them
    _navi_reached = None              # mask_navi_reached as disable_navi left it (dynamics.py:185-204)

    def init(self, tl_state: Tensor, gt_valid: Tensor, gt_pose: Tensor, gt_motion: Tensor, ag_type: Tensor, ag_attr: Tensor,
             ag_latent: Optional[Tensor], ag_latent_valid: Optional[Tensor], ag_navi: Optional[Tensor], ag_navi_valid: Tensor,
             **kwargs) -> None:
        """The reference's `self.dynamics.init(tl_state=tl_state_gt, **ag_tokens)` (waymo_motion.py:228). The simulation state itself
        lives in a rollout engine's device buffers; it is built from these arguments by the first `WaymoMotion.forward` that follows
        (which brings the map / light tokens the engine also needs). Until then the state attributes read step 0 of the ground truth."""
        if ag_latent is None or ag_navi is None:
            raise NotImplementedError("Dynamics.init: the default configuration has a latent and a destination per agent")
        self._pending = dict(tl_state_gt=tl_state, gt_valid=gt_valid, gt_pose=gt_pose, gt_motion=gt_motion, ag_type=ag_type, ag_attr=ag_attr,
                             ag_latent=ag_latent, ag_latent_valid=ag_latent_valid, ag_navi=ag_navi, ag_navi_valid=ag_navi_valid,
                             ag_navi_log_prob=kwargs.get("ag_navi_log_prob"))
        self._eng, self._navi_reached = None, None
        self.ag_navi_updated = True

    # ------------------------------------------------------------------ state of the bound rollout engine (read-only views)
    def bind(self, engine) -> None:
        self._eng, self._pending, self._navi_reached = engine, None, None

    def _s(self, key: str) -> Tensor:
        if self._eng is None and self._pending is not None:  # between init() and the first forward: step 0 of the ground truth
            import torch

            p = self._pending
            first = {"ag_valid": lambda: p["gt_valid"][:, :, 0].to(torch.uint8), "ag_disabled": lambda: torch.zeros_like(p["gt_valid"][:, :, 0], dtype=torch.uint8),
                     "ag_pose": lambda: p["gt_pose"][:, :, 0], "ag_motion": lambda: p["gt_motion"][:, :, 0],
                     "navi_valid": lambda: p["ag_navi_valid"].to(torch.uint8), "now_reached": lambda: torch.zeros_like(p["ag_navi_valid"], dtype=torch.uint8)}
            if key in first:
                return first[key]()
        if self._eng is None:
            raise RuntimeError("no rollout in progress: WaymoMotion.rollout / begin_rollout / Dynamics.init binds the simulation state")
        if key not in self._eng.S:
            raise RuntimeError(f"Dynamics.{key}: only a step-wise rollout keeps this per-step state "
                               "(WaymoMotion.rollout(..., stepwise=True) / begin_rollout(..., stepwise=True))")
        return self._eng.S[key]

    ag_valid = property(lambda self: self._s("ag_valid").bool())
    ag_disabled = property(lambda self: self._s("ag_disabled").bool())
    ag_pose = property(lambda self: self._s("ag_pose"))
    ag_motion = property(lambda self: self._s("ag_motion"))
    ag_navi_valid = property(lambda self: self._s("navi_valid").bool())
    mask_navi_reached = property(lambda self: self._navi_reached if self._navi_reached is not None else self._s("now_reached").bool())
    ag_navi = property(lambda self: self._pending["ag_navi"] if self._eng is None and self._pending is not None else self._eng.dest)
    ag_type = property(lambda self: self._pending["ag_type"] if self._eng is None and self._pending is not None else self._eng.ag_type)

    @property
    def tl_state(self) -> Tensor:
        """[n_sc, n_tl, 5] one-hot bool (the engine keeps a 5-bit mask per light)."""
        import torch

        if self._eng is None and self._pending is not None:
            return self._pending["tl_state_gt"][:, :, 0]
        m = self._s("tl_state")
        return ((m.to(torch.int32).unsqueeze(-1) >> torch.arange(5, device=m.device, dtype=torch.int32)) & 1).bool()

    def _stepwise(self) -> None:
        if self._eng is None or not self._eng.stepwise:
            raise RuntimeError("disable_ag / disable_navi act on a step-wise rollout (the engine's own loop applies them on the device)")

    def disable_ag(self, traffic_rule_violation: Dict[str, Tensor], gt_valid: Optional[Tensor] = None) -> None:
        """dynamics.py:165-183."""
        self._stepwise()
        self._eng.disable(outside=traffic_rule_violation["outside_map_this_step"], gt_valid=gt_valid)

    def disable_navi(self, traffic_rule_violation: Dict[str, Tensor]) -> None:
        """dynamics.py:185-204 (navi_mode dest)."""
        self._stepwise()
        self._navi_reached = traffic_rule_violation["dest_reached_this_step"]
        self._eng.disable(reached=self._navi_reached)
