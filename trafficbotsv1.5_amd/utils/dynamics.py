"""`Dynamics` / `MultiPathPP` parameter holders (utils/dynamics.py:13-274). The state update itself
(tanh-bounded action -> MultiPath++ midpoint integration -> overrides -> disabling) is `tbx_sim_step`."""
from typing import Tuple


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
