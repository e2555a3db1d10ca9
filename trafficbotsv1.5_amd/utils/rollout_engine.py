"""Closed-loop rollout engine: the device-resident simulation state + one hipGraph that replays
[policy (tl encoder -> agent encoder -> heads) -> tbx_sim_step] once per 0.1 s step.

The traffic lights' recurrence (window -> tl encoder -> logits -> next state) never reads an agent, so it runs one step
ahead on a second HIP stream: a replay of step t forks into { lights(t), tl encoder of the new window } and
{ agent policy on the tl K/V tables of step t-1's window, agents(t) }, joins and bumps the step counter. Results are
those of the sequential order (same inputs to every kernel); with one 64-agent scene the two halves occupy 8 and 4
workgroups of a 256-CU device, so they overlap completely.

Replaces the Python loop of `WaymoMotion.rollout` (pl_modules/waymo_motion.py:206-311) and everything it drives per
step (Dynamics, TeacherForcing.get, the feeding-back rule checks, RolloutBuffer.add, TrafficBots._append_hist); the
reference's >= 19 host synchronisations per step (SURVEY.md Appx D.1) are gone: the step index lives on the device.
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from .. import engine, hip
from .buffer import RolloutBuffer
from .traffic_rule_checker import dest_tables


def _u8(t: Tensor) -> Tensor:
    return t.to(torch.uint8).contiguous()


def _state_bits(one_hot: Tensor) -> Tensor:
    """[..., 5] one-hot bool -> u8 bit mask."""
    w = (1 << torch.arange(one_hot.shape[-1], device=one_hot.device, dtype=torch.int32))
    return (one_hot.to(torch.int32) * w).sum(-1).to(torch.uint8).contiguous()


def _scheduled(fn):
    """Run an engine method under the engine's own schedule (engine.use) and without autograd."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with engine.use(self.sched), torch.no_grad():
            return fn(self, *a, **kw)

    return wrapped


def lights_per_scene(tl_tokens: Dict[str, Tensor], k: int) -> Dict[str, Tensor]:
    """Per-scene view of traffic-light tokens that were expanded per rollout (`encode_scene(n_rollout=k)`: every per-light
    tensor repeated k times along the batch, map targets indexed per scene through mp_batch_div = k): entry 0 of every group
    of k, map targets indexed directly."""
    n = tl_tokens["tl_token_pose"].shape[0]
    out = {}
    for key, v in tl_tokens.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n and key != "mp_feat_flat":
            out[key] = v[::k].contiguous()
        elif not key.startswith("_"):  # caches (K/V tables keyed by module) are rebuilt for the view
            out[key] = v
    out.update(mp_batch_div=1, tl_batch_div=k, ag_mp_batch_div=k)
    return out


class RolloutEngine:
    """`schedule` (engine.Schedule; default: the caller's current one) is fixed at construction and made current for every call
    into the engine: two engines with different schedules (fp32 / bf16 tables, one / two streams) coexist in one process."""
    _streams: Dict[int, tuple] = {}

    @classmethod
    def _side_streams(cls, dev):
        """One (lights, K-nearest) stream pair per device for every engine of the process: engines of one process run one
        after the other, and torch hands out its 32 pool streams round-robin, so a fresh pair per rollout would soon share
        streams with unrelated work."""
        key = torch.device(dev).index or 0
        if key not in cls._streams:
            cls._streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return cls._streams[key]

    def __init__(self, model, dynamics, device, schedule: Optional[engine.Schedule] = None) -> None:
        self.model, self.dyn, self.dev = model, dynamics, device
        self.sched = schedule if schedule is not None else engine.current()
        self._graph_steps = 1  # steps per replay of graph_multi, fixed when it is captured
        self.reused = False    # True: kept by its owner across rollouts (refill): buffer() hands out copies of the logs
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graph_multi: Optional[Dict[int, torch.cuda.CUDAGraph]] = None  # by starting parity

    # ------------------------------------------------------------------ setup
    @_scheduled
    def reset(self, *, gt_valid: Tensor, gt_pose: Tensor, gt_motion: Tensor, tl_state_gt: Tensor, tf_mask: Tensor,
              ag_type: Tensor, ag_attr: Tensor, ag_latent: Tensor, ag_latent_valid: Tensor, ag_navi: Tensor,
              ag_navi_valid: Tensor, mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor], map_valid: Optional[Tensor],
              map_type: Optional[Tensor], map_pos: Optional[Tensor], map_dir: Optional[Tensor], map_boundary: Optional[Tensor], n_step: int,
              reward_weights=(0.1, 10.0, 0.1), ag_navi_log_prob: Optional[Tensor] = None, stepwise: bool = False,
              _lights_ahead_pass: bool = True, _tl_div: Optional[int] = None) -> None:
        """All tensors on the device. gt_* [n,A,Tg(,3)], tl_state_gt [n,L,Tt,5] bool, tf_mask [n,A,Tg] bool
        (TeacherForcing.ag_teacher_forcing), ag_navi [n,A] int64 dest; map_* are the raw polylines of the scene(s)
        ([n/div, M, N, ..]) for the destination check. reward_weights = (l_pos, l_rot, l_spd).weight of the
        differentiable reward logged per step. stepwise: the engine is driven through `forward_step` (WaymoMotion.forward)
        with explicit overrides instead of `run`."""
        dev = self.dev
        self._shape_key = self.shape_key(gt_valid=gt_valid, tl_state_gt=tl_state_gt, map_valid=map_valid, n_step=n_step, stepwise=stepwise,
                                         mp_tokens=mp_tokens, tl_tokens=tl_tokens, ag_latent=ag_latent)
        n, A, Tg = gt_valid.shape
        L, Tt = tl_state_gt.shape[1], tl_state_gt.shape[2]
        W = self.model.temp_window_size
        N = map_valid.shape[2] if map_valid is not None else 1
        div = tl_tokens.get("mp_batch_div", 1)
        f32, u8 = torch.float32, torch.uint8
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=dev)
        self.n, self.A, self.L, self.T, self.W = n, A, L, n_step, W
        # lights once per scene when the K = div rollouts of every scene were given identical lights (one host check per reset)
        kl = 1
        self.tl_share_ok = None
        if self.sched.share_lights and not stepwise and div > 1 and n % div == 0 and tl_tokens.get("tl_batch_div", 1) == 1:
            g = tl_state_gt.reshape(n // div, div, *tl_state_gt.shape[1:])
            tv = tl_tokens["tl_token_valid"].reshape(n // div, div, L)
            tp = tl_tokens["tl_token_pose"].reshape(n // div, div, L, 3)
            same = (g == g[:, :1]).all() & (tv == tv[:, :1]).all() & (tp == tp[:, :1]).all()
            if _tl_div is None:
                if bool(same):  # (one host check per reset)
                    kl = div
            else:
                # refill: the engine's graphs were captured for _tl_div; the new scene's lights must be shared the same way. The
                # check stays on the device (no host round trip between scenes) and is read when the log is handed out (buffer())
                # (an engine captured with unshared lights runs a scene whose rollouts DO share them correctly, only without the saving:
                # nothing to check then)
                kl = _tl_div
                self.tl_share_ok = same if kl > 1 else None
        elif _tl_div is not None:
            kl = _tl_div
        self.tl_div = kl
        nl = n // kl  # light batch entries
        self.tl_invalid_full = tl_tokens["tl_token_invalid"] if "tl_token_invalid" in tl_tokens else ~tl_tokens["tl_token_valid"]
        if kl > 1:
            tl_tokens = lights_per_scene(tl_tokens, kl)
            tl_state_gt = tl_state_gt[::kl]
        S = {}
        # ---- static
        S["ag_type_idx"] = _u8(ag_type.to(u8).argmax(-1))
        S["tf_mask"], S["gt_valid"] = _u8(tf_mask), _u8(gt_valid)
        S["gt_pose"], S["gt_motion"] = gt_pose.float().contiguous(), gt_motion.float().contiguous()
        S["tl_gt"] = _state_bits(tl_state_gt)
        if map_valid is None:
            # a step-wise engine built by WaymoMotion.forward after the reference's prologue (Dynamics.init): the caller's
            # TrafficRuleChecker.check evaluates outside-map / destination-reached itself (tbx_rule_navi_check) and hands the flags to
            # Dynamics.disable_ag / disable_navi - the copies inside tbx_sim_step get tables that never fire
            assert stepwise, "only a step-wise engine runs without the scene's polylines"
            N = 1
            S["boundary"] = torch.full((n, 4), math.inf, dtype=f32, device=dev)
            S["boundary"][:, 0::2] = -math.inf
            S["dest_pos"], S["dest_dir"], S["dest_invalid"] = z(n, A, 1, 2), z(n, A, 1, 2), torch.ones(n, A, 1, dtype=u8, device=dev)
            S["dest_kind"], S["dest_thresh"] = z(n, A, dt=u8), z(n, A)
        else:
            S["boundary"] = map_boundary.float().repeat_interleave(n // map_boundary.shape[0], 0).contiguous()
            dt_ = dest_tables(ag_navi, map_valid, map_type, map_pos, map_dir)  # traffic_rule_checker.py:87-107
            S["dest_pos"], S["dest_dir"], S["dest_invalid"] = dt_["pos"], dt_["dir"], dt_["invalid"]
            S["dest_kind"], S["dest_thresh"] = dt_["kind"], dt_["thresh"]
        self.ag_attr6 = ag_attr.float().contiguous()
        self.ag_type, self.navi_log_prob0, self.navi_valid0 = ag_type, ag_navi_log_prob, ag_navi_valid
        self.n_step_tl_gt, self.stepwise = Tt, stepwise
        self.ag_latent = ag_latent.reshape(n * A, -1).float().contiguous()
        self.latent_invalid = _u8(~ag_latent_valid.reshape(-1))
        self.dest = ag_navi.contiguous()
        # ---- initial dynamic state (Dynamics.init, dynamics.py:29-64) and its pristine copy
        init = dict(
            # [step index = 1, arrival counter of the fused advance = 0], built on the device: torch.tensor([..], device=) is a
            # BLOCKING host-to-device copy - it would wait for the previous scene's whole rollout on this stream (refill)
            step=torch.arange(1, -1, -1, dtype=torch.int32, device=dev),
            step_tl=torch.arange(1, -1, -1, dtype=torch.int32, device=dev),  # the lights' own copy (sim_state_tl_own below)
            ag_valid=_u8(gt_valid[:, :, 0]), ag_disabled=z(n, A, dt=u8), ag_pose=gt_pose[:, :, 0].float().contiguous(),
            ag_motion=gt_motion[:, :, 0].float().contiguous(), navi_valid=_u8(ag_navi_valid), outside_map=z(n, A, dt=u8),
            dest_reached=z(n, A, dt=u8), tl_state=S["tl_gt"][:, :, 0].contiguous(),
            hist_valid=z(n, A, W, dt=u8), hist_pose=z(n, A, W, 3), hist_motion=z(n, A, W, 3),
            hist_tl=torch.full((nl, L, W), 0xFF, dtype=u8, device=dev))
        init["hist_valid"][:, :, -1] = init["ag_valid"]
        init["hist_pose"][:, :, -1] = init["ag_pose"]
        init["hist_motion"][:, :, -1] = init["ag_motion"]
        init["hist_tl"][:, :, -1] = init["tl_state"]
        self.init_state = init
        for k, v in init.items():
            S[k] = v.clone()
        # ---- per-step model outputs and the rollout log
        S["action_mean"], S["tl_logits"] = z(n * A, 2), z(nl * L, 5)
        S.update(out_valid=z(n, A, n_step, dt=u8), out_pose=z(n, A, n_step, 3), out_motion=z(n, A, n_step, 3),
                 out_action=z(n, A, n_step, 2), out_tl_state=z(nl, L, n_step, dt=u8), out_outside_map=z(n, A, n_step, dt=u8),
                 out_dest_reached=z(n, A, n_step, dt=u8),
                 # the rest of RolloutBuffer.add (buffer.py:39-78): reward terms, its validity, the forcing mask, light NLL
                 out_reward=z(n, A, n_step, 4), out_reward_valid=z(n, A, n_step, dt=u8), out_tf=z(n, A, n_step, dt=u8),
                 out_tl_nll=z(nl, L, n_step))
        if stepwise:  # WaymoMotion.forward: this step's overrides, the player's actions, the this-step rule flags
            S.update(ov_valid=z(n, A, dt=u8), ov_pose=z(n, A, 3), ov_motion=z(n, A, 3), ov_tl_valid=z(nl, L, dt=u8),
                     ov_tl_state=z(nl, L, dt=u8), now_outside=z(n, A, dt=u8), now_reached=z(n, A, dt=u8),
                     player_valid=z(n, A, dt=u8), player_action=z(n, A, 2))
        self.S = S
        self.mp_tokens, self.tl_tokens = mp_tokens, tl_tokens
        st = hip.SimState()
        st.n_batch, st.n_ag, st.n_tl, st.window = n, A, L, W
        st.n_step_gt, st.n_step_tl_gt, st.n_step_out, st.n_node = Tg, Tt, n_step, N
        for name, _ in hip.SimState._fields_:
            if name in S:
                setattr(st, name, S[name].data_ptr())
        st.max_acc = (C.c_float * 3)(*self.dyn.max_acc)
        st.max_yaw_rate = (C.c_float * 3)(*self.dyn.max_yaw_rate)
        st.dt = self.dyn.dt
        st.w_pos, st.w_rot, st.w_spd = (float(w) for w in reward_weights)
        self.sim_state = st
        # the lights' part of tbx_sim_step only touches the light arrays: its own descriptor with their batch size
        self.sim_state_tl = st
        if kl > 1:
            stl = hip.SimState()
            C.memmove(C.byref(stl), C.byref(st), C.sizeof(hip.SimState))
            stl.n_batch = nl
            self.sim_state_tl = stl
        # Schedule.sim_before_join: the lights' stream counts its steps itself (TBX_SIM_LIGHTS | TBX_SIM_ADVANCE on `step_tl`) - with one
        # shared counter the agents' advance has to be ordered behind the lights' read of it, a cross-queue wait (~6 us of idle
        # queue even when long since satisfied) in front of the launch that closes every step
        own = hip.SimState()
        C.memmove(C.byref(own), C.byref(self.sim_state_tl), C.sizeof(hip.SimState))
        own.step = S["step_tl"].data_ptr()
        self.sim_state_tl_own = own
        self.policy_out = dict(action_mean=S["action_mean"], tl_logits=S["tl_logits"])
        self.graph = self.graph_multi = self.graph_prime = self.graph_refill = None
        self._prime_prepares = False
        self._tl_prep = None
        self.one_queue = self._one_queue_ok(n, A, nl, L, stepwise)
        self.tl_kv = None  # two K/V table buffers of the light tokens: agents of step t read [t & 1], the lights' pass writes the other
        self.parity = 0
        self._prep_ready = False  # this step's tbx_agent_prep already ran in the previous step's fused tail (step())
        self.side, self.aux = self._side_streams(dev)
        # per-rollout constants of the heads chain (embedded latent, destination feature): once, not every step
        self.consts = (self.model.rollout_constants(self.ag_latent, self.dest, mp_tokens, div, latent_invalid=self.latent_invalid)
                       if self.sched.hoist_constants else None)
        self._n_forward = 0
        if not stepwise and _lights_ahead_pass:
            self._tl_ahead(0)

    @staticmethod
    def shape_key(*, gt_valid, tl_state_gt, map_valid, n_step, mp_tokens, tl_tokens, ag_latent, stepwise=False, **_):
        """What must agree for `refill` (same buffers, same captured graphs): every shape the engine's buffers depend on."""
        tok = lambda d: tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in d.items() if torch.is_tensor(v))) + tuple(
            sorted((k, v) for k, v in d.items() if isinstance(v, int)))
        return (tuple(gt_valid.shape), tuple(tl_state_gt.shape), None if map_valid is None else tuple(map_valid.shape), int(n_step), bool(stepwise), tuple(ag_latent.shape),
                tok(mp_tokens), tok(tl_tokens))

    @_scheduled
    def refill(self, **kw) -> None:
        """ANOTHER scene of the same shapes (reset's arguments) into this engine's buffers, in place: every pointer the captured
        hipGraphs hold stays valid, so a loop over scenes (the reference's validation_step, waymo_motion.py:526) pays the capture
        once per shape instead of once per rollout. The engine owns the token dicts it was first reset with: their tensors are
        overwritten. Results are those of a fresh engine, bit for bit (tests/test_hip_rollout.py).
        = refill_prepare (reads nothing of this engine's state: may run beside its rollout) + refill_commit."""
        self.refill_commit(self.refill_prepare.__wrapped__(self, **kw))

    @_scheduled
    def refill_prepare(self, **kw) -> "RolloutEngine":
        """The derived state of the new scene as a scratch engine (reset's arithmetic on reset's arguments). Touches none of THIS
        engine's buffers: it may be enqueued on another stream while this engine's rollout of the previous scene is still running."""
        key = self.shape_key(**kw)
        if key != self._shape_key:
            raise ValueError("refill: shapes differ from the ones this engine was built for (use a new engine)")
        fresh = RolloutEngine(self.model, self.dyn, self.dev, schedule=self.sched)
        RolloutEngine.reset.__wrapped__(fresh, _lights_ahead_pass=False, _tl_div=self.tl_div, **kw)
        return fresh

    @_scheduled
    def refill_commit(self, fresh: "RolloutEngine") -> None:
        """refill_prepare's result into this engine's buffers (the ones its graphs read) + the map K/V tables + the priming of the
        first step. Must follow this engine's previous rollout in stream order."""
        assert fresh.S.keys() == self.S.keys()
        # every copy of the commit as a few multi-tensor launches (_copy_all) instead of ~80 single ones: 4-8 us each on the stream
        # the next rollout waits on
        pairs: List[Tuple[Tensor, Tensor]] = []
        for k, v in fresh.S.items():
            pairs.append((self.S[k], v))
        for k, v in fresh.init_state.items():
            pairs.append((self.init_state[k], v))
        for name in ("ag_attr6", "ag_latent", "latent_invalid", "dest"):
            pairs.append((getattr(self, name), getattr(fresh, name)))
        # (what buffer() reads on the host side: engine-owned copies - `fresh` may be overwritten by the NEXT scene's prepare while
        # this scene's log is still to be handed out)
        if torch.is_tensor(self.tl_invalid_full):
            self.tl_invalid_full = self._own("tl_invalid_full", fresh.tl_invalid_full, pairs)
        # the device flag of the light-sharing check (read in buffer(): no host round trip between scenes) is engine-owned for the same
        # reason - `fresh` lives in graph_prepare's pool and the NEXT scene's prefetch rewrites it on the side stream before this
        # scene's buffer() reads it (the flag read would be the next scene's, or race with its write)
        self.tl_share_ok = None if fresh.tl_share_ok is None else self._own("tl_share_ok", fresh.tl_share_ok, pairs)
        self.ag_type = self._own("ag_type", fresh.ag_type, pairs)
        self.navi_valid0 = self._own("navi_valid0", fresh.navi_valid0, pairs)
        self.navi_log_prob0 = None if fresh.navi_log_prob0 is None else self._own("navi_log_prob0", fresh.navi_log_prob0, pairs)
        late = self._copy_tokens(self.mp_tokens, fresh.mp_tokens, pairs) + self._copy_tokens(self.tl_tokens, fresh.tl_tokens, pairs)
        if self.consts is not None:
            for k, v in fresh.consts.items():
                if torch.is_tensor(v):
                    pairs.append((self.consts[k], v))
                else:
                    assert self.consts[k] == v
        self._copy_all(pairs)
        for dst, src in late:  # (u8 views of masks refreshed above)
            dst.copy_(src.to(torch.uint8))
        # per-scene K/V tables of the map tokens (cached in the token dicts): recomputed into the tables the graphs read
        self.model.ag_encoder.kv_mp(self.mp_tokens, refresh=True)
        self.model.tl_encoder._kv_mp(self.tl_tokens, refresh=True)
        self.parity = 0
        self._n_forward = 0
        self._prime()

    def _own(self, name: str, src: Tensor, pairs: Optional[list] = None) -> Tensor:
        """An engine-owned tensor `name` holding a copy of src (allocated once, refilled in place; pairs: the copy is queued there)."""
        own = self.__dict__.setdefault("_owned", {})
        if name not in own or own[name].shape != src.shape or own[name].dtype != src.dtype:
            own[name] = torch.empty_like(src)
        if pairs is None:
            own[name].copy_(src)
        else:
            pairs.append((own[name], src))
        return own[name]

    @staticmethod
    def _copy_all(pairs) -> None:
        """dst.copy_(src) for every pair; same-dtype contiguous pairs of equal shape go through torch._foreach_copy_ (one launch per
        few dozen tensors), the rest one by one."""
        groups: Dict[torch.dtype, Tuple[list, list]] = {}
        for dst, src in pairs:
            if dst.data_ptr() == src.data_ptr():
                continue
            if dst.dtype == src.dtype and dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous() and dst.device == src.device:
                g = groups.setdefault(dst.dtype, ([], []))
                g[0].append(dst)
                g[1].append(src)
            else:
                dst.copy_(src)
        for dsts, srcs in groups.values():
            torch._foreach_copy_(dsts, srcs)

    def capture_refill(self, make_kw) -> None:
        """`refill` of this engine as TWO hipGraphs. make_kw() builds reset's keyword arguments from STATIC input tensors (the caller
        overwrites those in place, then replays):
          graph_prepare  make_kw() + refill_prepare: the once-per-scene encoders if make_kw runs them + the derived state. Reads
                         nothing of the engine: replayed on a SIDE stream beside the previous scene's rollout (prefetch_refill);
          graph_commit   refill_commit: ~80 copies into the buffers the step graphs read, the map K/V tables, the lights' first pass,
                         the first tbx_agent_prep. Replayed on the launch stream behind that rollout (commit_refill).
        ~300 launch-bound launches per scene become two (a loop over scenes paid ~2 ms of device time per scene for them eagerly,
        tools/scene_loop_profile.py). Nothing in refill waits for the device or draws random numbers, so the capture is exact:
        replays equal eager refills bit for bit (tests/test_hip_boundary.py)."""
        assert self.graph is not None, "capture() the step graphs first"
        main = torch.cuda.current_stream()
        self._refill_side = torch.cuda.Stream(device=self.dev)
        side = self._refill_side
        side.wait_stream(main)
        with torch.cuda.stream(side):  # eager warm-up off the launch stream (allocator growth, weight images, caches)
            self.refill(**make_kw())
        main.wait_stream(side)
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            fresh = self.refill_prepare(**make_kw())
        with torch.cuda.graph(gb):
            self.refill_commit(fresh)
        self.graph_prepare, self.graph_commit, self._fresh_static = ga, gb, fresh
        self.graph_refill = gb
        self._tl_share_static = self.tl_share_ok  # (the engine-owned device flag of the light-sharing check, rewritten by every graph_commit replay)
        self._ev_prepared = self._ev_committed = None

    def prefetch_refill(self, fill_inputs=None) -> None:
        """graph_prepare on the side stream: fill_inputs() (the caller's in-place copies into the static inputs) + the replay. Call
        it right after the previous scene's run() was enqueued - it executes beside that rollout."""
        main, side = torch.cuda.current_stream(), self._refill_side
        side.wait_stream(main)  # (the scene tensors were made on the launch stream; the previous commit has read `fresh`)
        with torch.cuda.stream(side):
            if fill_inputs is not None:
                fill_inputs()
            self.graph_prepare.replay()
            self._ev_prepared = side.record_event()

    def commit_refill(self) -> None:
        """graph_commit on the launch stream, behind the prepare and behind this engine's previous rollout (stream order)."""
        main = torch.cuda.current_stream()
        if self._ev_prepared is not None:
            main.wait_event(self._ev_prepared)
            self._ev_prepared = None
        self.graph_commit.replay()
        self.tl_share_ok = self._tl_share_static
        self.parity = 0
        self._n_forward = 0
        self._prep_ready = self._prime_prepares

    def replay_refill(self, fill_inputs=None) -> None:
        """prefetch_refill + commit_refill back to back (no overlap with a rollout)."""
        self.prefetch_refill(fill_inputs)
        self.commit_refill()

    @staticmethod
    def _copy_tokens(dst: Dict[str, Tensor], src: Dict[str, Tensor], pairs: list) -> list:
        """Queues the token tensors' copies in `pairs`; returns the (u8 copy, its source mask) pairs to refresh AFTER those ran."""
        for k, v in src.items():
            if k.startswith("_"):
                continue  # caches: rebuilt by their owners
            if torch.is_tensor(v):
                if dst[k].data_ptr() != v.data_ptr():
                    pairs.append((dst[k], v))
            elif isinstance(v, (int, float, bool)):
                assert dst[k] == v, (k, dst[k], v)
        late = []
        for k in dst:  # u8 copies of a mask that a consumer added lazily (e.g. mp_token_invalid_u8): refreshed from their source
            if k.endswith("_u8") and k not in src and torch.is_tensor(dst.get(k[:-3])):
                late.append((dst[k], dst[k[:-3]]))
        return late

    @_scheduled
    def restore(self) -> None:
        """Back to step 1 without re-allocating (pointers captured in the graph stay valid)."""
        self._copy_all([(self.S[k], v) for k, v in self.init_state.items()])
        self.parity = 0
        self._n_forward = 0
        self._prime()

    def _prime(self) -> None:
        """What a (re)set state needs before its first step: the lights' encoder pass on the initial window (tables for the agents'
        first step, logits for the lights' first update) and, with a fused tail, the first step's tbx_agent_prep. Eagerly these are ~7
        launches behind ~1.2 ms of host work per rollout; capture() records them once (graph_prime: every buffer they touch is
        refilled in place) and every later restore() / refill() replays that."""
        if self.graph_prime is not None and not torch.cuda.is_current_stream_capturing():  # (inside capture_refill: eagerly, into that graph)
            self.policy_out["tl_kv"] = self.tl_kv[0]
            self.graph_prime.replay()
            self._prep_ready = self._prime_prepares
            return
        if not self.stepwise:
            self._tl_ahead(0)
        self._prime_prep()

    def _prime_prep(self) -> None:
        """Schedule.fused_tail: a step's tbx_agent_prep is run by the previous step's last launch - for the first step after the state
        was (re)set it is run here, eagerly, so that every captured step starts from prepared windows."""
        prep = self.policy_out.get("prep")
        self._prep_ready = False
        if prep is None or not prep.get("_tail_fused") or self.stepwise:
            return  # (no step has run yet: the first one prepares its windows itself; or this schedule does not fuse the tail)
        S = self.S
        div = self.tl_tokens.get("ag_mp_batch_div", self.tl_tokens.get("mp_batch_div", 1))
        self.model.ag_encoder.run_prep(S["hist_valid"], S["hist_pose"], S["hist_motion"], self.ag_attr6, S["ag_type_idx"], prep, self.dest,
                                       self.mp_tokens, div)
        self._prep_ready = True

    def _tl_ahead(self, slot: int, prepared=None) -> None:
        """tl encoder on the current light window: logits for the next lights update, K/V tables (into buffer `slot`)
        for the next agent step. prepared: the window's tbx_tl_prep rows were written by the lights' update (step()).
        One-queue engines (self.one_queue): the pass ENDS with the lights' update (tbx_tl_tail_t.sim_state) - the window it leaves is the
        next pass's, its tbx_tl_prep rows included."""
        sim = self._tl_tail_sim() if self.one_queue else None
        if self.tl_kv is None:
            self.policy_out.pop("tl_kv", None)
            kv = self.model.tl_policy(self.S["hist_tl"], self.tl_tokens, self.policy_out, tail_sim=sim)
            self.tl_kv = [kv, torch.empty_like(kv)] if slot == 0 else [torch.empty_like(kv), kv]
            return
        self.policy_out["tl_kv"] = self.tl_kv[slot]
        self.model.tl_policy(self.S["hist_tl"], self.tl_tokens, self.policy_out, prepared=prepared, tail_sim=sim)

    def _tl_tail_sim(self) -> dict:
        if self._tl_prep is None:
            hist = self.S["hist_tl"]
            self._tl_prep = self.model.tl_encoder.prep_buffers(hist.shape[0], hist.shape[1], hist.device)
        return dict(state=self.sim_state_tl_own, parts=hip.SIM_LIGHTS | hip.SIM_ADVANCE, attr=self._tl_prep[0], row_invalid=self._tl_prep[1])

    def _one_queue_ok(self, n: int, A: int, nl: int, L: int, stepwise: bool) -> bool:
        """The one-queue step applies: both blocks on the one-launch decoder layer with their tails in the launch (Schedule docstring)."""
        c = self.sched
        return bool(c.one_queue and c.lights_ahead and c.sim_before_join and c.fused_tail and c.tl_prep_rides and c.dec_layer and c.dec_mid
                    and c.dec_tail_mfma and c.front_fused and c.tile_small and c.heads_tail and c.navi_rider and c.knn_main and not stepwise
                    and n * A <= min(c.live_max_agents, c.live_max) and nl * L <= c.live_max and self.model.tl_encoder.temp_window_size <= 16)

    # ------------------------------------------------------------------ stepping
    @_scheduled
    def step(self) -> None:
        S, main = self.S, torch.cuda.current_stream()
        if not self.sched.lights_ahead:
            self.model.policy_step(S["hist_valid"], S["hist_pose"], S["hist_motion"], S["hist_tl"], self.ag_attr6,
                                   S["ag_type_idx"], self.ag_latent, self.latent_invalid, self.dest, S["navi_valid"],
                                   self.tl_tokens, self.mp_tokens, self.policy_out, rollout_consts=self.consts)
            if self.tl_div > 1:
                hip.sim_step(self.sim_state_tl, hip.SIM_LIGHTS)
                hip.sim_step(self.sim_state, hip.SIM_AGENTS | hip.SIM_ADVANCE)
            else:
                hip.sim_step(self.sim_state)
            return
        if self.one_queue:
            return self._step_one_queue()
        p = self.parity
        early = self.sched.sim_before_join
        st_tl, parts_tl = (self.sim_state_tl_own, hip.SIM_LIGHTS | hip.SIM_ADVANCE) if early else (self.sim_state_tl, hip.SIM_LIGHTS)
        if early:  # (both counters advance once per step; a caller that mixed this with the one-stream forms would desynchronise them)
            assert not self.stepwise
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            # logits of the previous tl encoder pass -> lights of this step (+ the one-hot rows of their new windows: tbx_sim_step_tl_prep)
            if self.sched.tl_prep_rides:
                if self._tl_prep is None:
                    hist = self.S["hist_tl"]
                    self._tl_prep = self.model.tl_encoder.prep_buffers(hist.shape[0], hist.shape[1], hist.device)
                hip.sim_step(st_tl, parts_tl, tl_prep=(self.tl_tokens["tl_token_invalid_u8"], *self._tl_prep))
                self._tl_ahead(1 - p, prepared=self._tl_prep)
            else:
                hip.sim_step(st_tl, parts_tl)
                self._tl_ahead(1 - p)
        fuse = early and self.sched.fused_tail
        self.model.agent_policy(S["hist_valid"], S["hist_pose"], S["hist_motion"], self.ag_attr6, S["ag_type_idx"],
                                self.ag_latent, self.latent_invalid, self.dest, S["navi_valid"], self.tl_tokens,
                                self.mp_tokens, self.tl_kv[p], self.policy_out, aux_stream=self.aux, rollout_consts=self.consts,
                                fused_tail=dict(sim_state=self.sim_state, parts=hip.SIM_AGENTS | hip.SIM_ADVANCE) if fuse else None,
                                prep_ready=fuse and self._prep_ready)
        if fuse and self.policy_out["prep"].get("_tail_fused"):
            # the last layer's launch ran the agents' step and the next step's tbx_agent_prep (tbx_heads_tail_t.sim_state / next_prep)
            self._prep_ready = True
            main.wait_stream(self.side)
        elif early:
            self._prep_ready = False
            # the agents' launch closes their step on their own stream; the join moves behind it (hipGraph's executor puts a node
            # on the queue of the parent it reaches first: joined first, this launch ran on the lights' queue behind a ~12 us wait)
            hip.sim_step(self.sim_state, hip.SIM_AGENTS | hip.SIM_ADVANCE)
            main.wait_stream(self.side)
        else:
            main.wait_stream(self.side)
            hip.sim_step(self.sim_state, hip.SIM_AGENTS | hip.SIM_ADVANCE)
        self.parity = 1 - p

    def _step_one_queue(self) -> None:
        """One closed-loop step as FIVE launches on the stepping stream (Schedule.one_queue): the lights' pass for the NEXT agent step
        (window -> tables into the other buffer, logits, then - in its last layer's tail - the lights' own update) paired launch by
        launch with the agents' step on this step's tables. The wrappers collect both halves' launch descriptors (hip.defer) instead
        of launching; the halves are independent, so pairing them changes no result."""
        S, p = self.S, self.parity
        tail = dict(sim_state=self.sim_state, parts=hip.SIM_AGENTS | hip.SIM_ADVANCE)
        if not self._prep_ready:
            # the very first step of a fresh engine (no tbx_agent_prep buffers yet: _prime_prep): the same launches one after the other
            self._tl_ahead(1 - p, prepared=self._tl_prep)
            self.model.agent_policy(S["hist_valid"], S["hist_pose"], S["hist_motion"], self.ag_attr6, S["ag_type_idx"], self.ag_latent,
                                    self.latent_invalid, self.dest, S["navi_valid"], self.tl_tokens, self.mp_tokens, self.tl_kv[p], self.policy_out,
                                    aux_stream=self.aux, rollout_consts=self.consts, fused_tail=tail, prep_ready=False)
            if not self.policy_out["prep"].get("_tail_fused"):
                raise RuntimeError("one-queue step: the agents' last layer did not take the fused tail; use Schedule(one_queue=False)")
            self._prep_ready = True
            self.parity = 1 - p
            return
        with hip.defer() as lights:
            self._tl_ahead(1 - p, prepared=self._tl_prep)  # (the previous pass's tail - the prime's for the first step - wrote these rows)
        with hip.defer() as agents:
            self.model.agent_policy(S["hist_valid"], S["hist_pose"], S["hist_motion"], self.ag_attr6, S["ag_type_idx"], self.ag_latent,
                                    self.latent_invalid, self.dest, S["navi_valid"], self.tl_tokens, self.mp_tokens, self.tl_kv[p], self.policy_out,
                                    aux_stream=self.aux, rollout_consts=self.consts, fused_tail=tail, prep_ready=True)
        want = ["tbx_front"] + ["tbx_knarpe_dec_layer"] * (len(lights) - 1)
        if not (len(lights) == len(agents) >= 2 and [c[0] for c in lights] == want == [c[0] for c in agents]
                and self.policy_out["prep"].get("_tail_fused")):
            raise RuntimeError("one-queue step: the two halves did not reduce to [tbx_front, tbx_knarpe_dec_layer x layers] with fused tails "
                               f"(lights {[c[0] for c in lights]}, agents {[c[0] for c in agents]}); use Schedule(one_queue=False)")
        hip.launch_front_pair(agents[0], lights[0])
        for a_, l_ in zip(agents[1:], lights[1:]):
            hip.launch_dec_layer_pair(a_, l_)
        self._prep_ready = True
        self.parity = 1 - p

    @_scheduled
    def capture(self) -> None:
        """Warm up one eager step, restore, then capture the step into hipGraphs (one per parity of the light-table
        double buffer; capturing executes nothing, so the state stays at the restored start)."""
        self.step()
        torch.cuda.synchronize()
        self.restore()
        graphs = []
        for p in ((0, 1) if self.sched.lights_ahead else (0,)):
            self.parity = p
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            graphs.append(g)
        # ... and sched.graph_steps consecutive steps as one graph (starting at parity 0): a replay boundary costs ~9 us of idle device
        # (the next graph's first kernel starts that long after the last one's last), a kernel boundary inside a graph ~1 us
        self.graph_multi = None
        self._graph_steps = max(1, int(self.sched.graph_steps) // 2 * 2) if self.sched.graph_steps > 1 else 1
        if self._graph_steps > 1:
            # one per parity it can start at (an odd number of warm-up steps leaves the double buffer at parity 1: without its own
            # graph the run fell back to single-step replays around a shorter multi-step one - ~9 us of idle device per extra replay)
            self.graph_multi = {}
            for p in range(len(graphs)):
                self.parity = p
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(self._graph_steps):
                        self.step()
                self.graph_multi[p] = g
        self.parity = 0
        self.graph = graphs
        # the priming of a (re)set state as a graph too (_prime): the state is at the restored, primed start - capturing executes nothing
        if self.sched.prime_graph and not self.stepwise:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._tl_ahead(0)
                self._prime_prep()
            self._prime_prepares = self._prep_ready
            self.graph_prime = g

    @_scheduled
    def run(self, n_steps: Optional[int] = None, use_graph: bool = True) -> None:
        n_steps = self.T if n_steps is None else n_steps
        if use_graph and self.graph is None:
            self.capture()
        done = 0
        while done < n_steps:
            if not use_graph:
                self.step()
                done += 1
            elif self.graph_multi is not None and n_steps - done >= self._graph_steps:
                self.graph_multi[self.parity % len(self.graph)].replay()  # an even number of steps: the parity is back where it was
                done += self._graph_steps
            else:
                self.graph[self.parity % len(self.graph)].replay()
                self.parity = 1 - self.parity
                done += 1

    # ------------------------------------------------------------------ step-wise driving (WaymoMotion.forward)
    @_scheduled
    def forward_step(self, ag_override: Dict[str, Tensor], tl_override: Dict[str, Tensor],
                     player_override: Optional[Dict[str, Tensor]] = None) -> int:
        """One `WaymoMotion.forward` (waymo_motion.py:118-204): append the current state to the windows (from the second call on:
        traffic_bots.py:123-143 does it at the head of every model call), policy, Dynamics.update_ag with the player's actions,
        override_ag / override_tl with the overrides given. The caller owns what the reference's `rollout` does around it: rule
        checks, disable_ag / disable_navi (`disable`). Returns the 0-based log slot the step was written to."""
        assert self.stepwise, "reset(..., stepwise=True) first"
        S = self.S
        if self._n_forward > 0:
            hip.sim_step(self.sim_state, hip.SIM_APPEND)  # (step-wise engines never share lights: tl_div == 1)
        S["ov_valid"].copy_(ag_override["valid"])
        S["ov_pose"].copy_(ag_override["pose"])
        S["ov_motion"].copy_(ag_override["motion"])
        S["ov_tl_valid"].copy_(tl_override["valid"])
        S["ov_tl_state"].copy_(_state_bits(tl_override["state"]))
        if player_override is not None:
            S["player_valid"].copy_(player_override["valid"])
            S["player_action"].copy_(player_override["action"])
        else:
            S["player_valid"].zero_()
        self.model.policy_step(S["hist_valid"], S["hist_pose"], S["hist_motion"], S["hist_tl"], self.ag_attr6, S["ag_type_idx"],
                               self.ag_latent, self.latent_invalid, self.dest, S["navi_valid"], self.tl_tokens, self.mp_tokens,
                               self.policy_out, rollout_consts=self.consts)  # (latent and destinations are fixed by begin_rollout here too)
        slot = self._n_forward
        hip.sim_step(self.sim_state, hip.SIM_AGENTS | hip.SIM_LIGHTS | hip.SIM_ADVANCE | hip.SIM_NO_DISABLE | hip.SIM_NO_APPEND)
        self._n_forward += 1
        return slot

    @torch.no_grad()
    def disable(self, outside: Optional[Tensor] = None, reached: Optional[Tensor] = None, gt_valid: Optional[Tensor] = None) -> None:
        """Dynamics.disable_ag (outside = outside_map_this_step, gt_valid) / disable_navi (reached = dest_reached_this_step)
        (dynamics.py:165-204) on the device state, for step-wise drivers."""
        S = self.S
        if outside is not None:
            dis = outside.to(torch.uint8)
            if gt_valid is not None:
                dis = dis & (~gt_valid.bool()).to(torch.uint8)
            S["ag_disabled"] |= dis
            S["ag_valid"] &= 1 - dis
        if reached is not None:
            S["navi_valid"] &= 1 - reached.to(torch.uint8)

    # ------------------------------------------------------------------ results
    def action_log_prob(self, valid_u8: Tensor) -> Tensor:
        """Dynamics.update_ag's action_log_prob with deterministic actions (dynamics.py:87-91): log N(mean | mean, std) of the
        2-d action = -sum_d (log_std_d + log sqrt(2 pi)) per agent type, 0 for invalid agents. valid_u8 [n, A, ...]."""
        ls = torch.stack(list(self.model.action_head.log_std), 0).sum(-1)  # [3]
        per_type = -(ls + self.model.action_head.out_dim * 0.5 * math.log(2 * math.pi))
        lp = per_type[self.S["ag_type_idx"].long()]  # [n, A]
        while lp.dim() < valid_u8.dim():
            lp = lp.unsqueeze(-1)
        return lp * valid_u8.to(lp.dtype)

    def buffer(self, step_current: int = 10, rule_checker=None) -> RolloutBuffer:
        """The rollout log as the reference's RolloutBuffer. With a TrafficRuleChecker, the five metric-only rule checks
        (collided, collided_wosac, run_road_edge, run_red_light, passive: waymo_motion.py:250 in the reference's loop) are
        evaluated here for all steps at once from the device-resident log (tbx_rule_check over n x T frames)."""
        S, buf = self.S, RolloutBuffer(self.T, step_current)
        if getattr(self, "tl_share_ok", None) is not None:
            ok, self.tl_share_ok = self.tl_share_ok, None
            if not bool(ok):  # (the one host read of a refilled engine's rollout; the log is about to be read anyway)
                raise RuntimeError("refill: the new scene's lights are (not) shared across its rollouts unlike the scene this engine was "
                                   "captured for (use a new engine: WaymoMotion.engine_cache = 0)")
        if self.reused:  # the log tensors are rewritten by this engine's next rollout: the buffer gets its own copies
            S = {k: (v.clone() if k.startswith("out_") else v) for k, v in S.items()}
        # (THESE u8 logs hold 0 / 1 - tbx_sim_step stores C++ bools into out_valid / out_outside_map / out_dest_reached /
        # out_reward_valid / out_tf -: their bool form is a reinterpreting view, not a conversion launch. Bit-mask logs (out_tl_state's
        # 5-bit states, the rule flags) never go through b8.)
        def b8(t):
            assert t.dtype == torch.uint8
            if os.environ.get("TBX_DEBUG_BOOL_LOGS"):
                assert int(t.max()) <= 1, "a 0/1 log holds a bit mask"
            return t.view(torch.bool)
        buf.pred_valid, buf.pred_pose, buf.pred_motion = b8(S["out_valid"]), S["out_pose"], S["out_motion"]
        buf.violation = {"outside_map": b8(S["out_outside_map"]), "dest_reached": b8(S["out_dest_reached"])}
        rep = (lambda t: t) if self.tl_div == 1 else (lambda t: t.repeat_interleave(self.tl_div, 0))  # per rollout again
        out_tl = rep(S["out_tl_state"])
        if rule_checker is not None:
            buf.violation.update(rule_checker.check_log(S["out_valid"], S["out_pose"], S["out_motion"], out_tl))
        bits = (out_tl.to(torch.int32).unsqueeze(-1) >> torch.arange(5, device=self.dev, dtype=torch.int32)) & 1
        buf.vis_dict = {"action": S["out_action"], "tl_state": bits.bool()}
        # what the reference's loop adds per step besides the prediction (waymo_motion.py:250-300)
        r = S["out_reward"]
        buf.diffbar_reward = {"diffbar_reward_valid": b8(S["out_reward_valid"]), "diffbar_reward": r[..., 3],
                              "r_imitation_pos": r[..., 0], "r_imitation_rot": r[..., 1], "r_imitation_spd": r[..., 2],
                              "r_traffic_rule_approx": torch.zeros_like(r[..., 0])}
        buf.tl_state_nll = rep(S["out_tl_nll"])
        inv = self.tl_invalid_full.bool().unsqueeze(-1).expand(-1, -1, self.T).clone()
        inv[:, :, max(self.n_step_tl_gt - 1, 0):] = True  # steps past the light ground truth carry no NLL (waymo_motion.py:277-279)
        buf.tl_state_nll_invalid = inv
        buf.mask_teacher_forcing = b8(S["out_tf"])
        buf.action_log_prob = self.action_log_prob(S["out_valid"])
        lp0 = self.navi_log_prob0 if self.navi_log_prob0 is not None else torch.zeros(self.n, self.A, device=self.dev)
        buf.navi_log_prob, buf.navi_log_prob_valid = lp0.unsqueeze(-1), self.navi_valid0.bool().unsqueeze(-1)
        buf._finished = True
        return buf
