"""`WaymoMotion` (pl_modules/waymo_motion.py:29-524): the closed-loop driver with the reference's entry points
`forward / rollout / reactive_replay / joint_future_pred / training_step / configure_optimizers`.

The rollout is a device-resident state machine (utils/rollout_engine.py) replaying one hipGraph per step; scenes and
rollouts are independent, so multi-GPU inference shards them over ranks with no collective.
WOMD / WOSAC metrics, submission writers and video logging of the reference are out of scope.
"""
import dataclasses
from collections import OrderedDict
from typing import Dict, Optional

import os

import torch
from torch import Tensor, nn

from .. import engine
from ..config import AttrDict, to_attr
from ..data_modules.scene_centric import SceneCentricPreProcessing
from ..models.modules.distributions import MyDist
from ..models.traffic_bots import TrafficBots
from ..utils.buffer import RolloutBuffer
from ..utils.dynamics import Dynamics
from ..utils.rewards import DifferentiableReward
from ..utils.rollout_engine import RolloutEngine
from ..utils.teacher_forcing import TeacherForcing
from ..utils.traffic_rule_checker import TrafficRuleChecker

try:  # Lightning is not in the container image; the hot path only needs hparams / log / current_epoch
    from pytorch_lightning import LightningModule  # type: ignore
except Exception:  # pragma: no cover
    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch, self.global_rank, self.logged = 0, 0, {}

        def save_hyperparameters(self):
            """The caller's constructor arguments as `self.hparams` (attribute access), as Lightning does."""
            import inspect

            av = inspect.getargvalues(inspect.currentframe().f_back)
            hp = {k: av.locals[k] for k in av.args if k != "self"}
            if av.keywords:
                hp.update(av.locals[av.keywords])
            self.hparams = to_attr(hp)

        def log(self, k, v, **kw):
            self.logged[k] = v


def _strip_target(cfg):
    return AttrDict({k: v for k, v in dict(cfg).items() if k != "_target_"})


class WaymoMotion(LightningModule):
    def __init__(self, time_step_current: int, time_step_gt: int, time_step_end: int, time_step_sim_start: int,
                 hidden_dim: int, data_size, pre_processing, model, p_training_rollout_prior: float,
                 training_detach_model_input: bool, training_deterministic_action: bool, pred_navi_after_reached: bool,
                 differentiable_reward, dynamics, teacher_forcing_training, teacher_forcing_reactive_replay,
                 teacher_forcing_joint_future_pred, training_metrics, optimizer, lr_scheduler, lr_navi: float,
                 n_joint_future_wosac: int = 32, joint_future_pred_deterministic_k0: bool = False, **unused) -> None:
        super().__init__()
        if pred_navi_after_reached:
            raise NotImplementedError("pred_navi_after_reached=False is the default (no per-step host branch)")
        # The training step batches the decoder over TIME (train_graph.py): that is exact only because every step's model input is
        # detached (waymo_motion.py:158-161) and the action is the distribution's mean (:370, dynamics.py:87-91). With either flag off
        # the reference back-propagates through the closed loop / samples actions - gradients this implementation does not compute.
        if not training_detach_model_input:
            raise NotImplementedError("training_detach_model_input=False: back-propagation through the closed loop is not implemented "
                                      "(the time-batched training step needs detached model inputs, the default)")
        if not training_deterministic_action:
            raise NotImplementedError("training_deterministic_action=False: sampled actions in the training rollout are not implemented "
                                      "(the default is the deterministic action)")
        self.save_hyperparameters()  # self.hparams.<constructor argument>, as in the reference (waymo_motion.py:66)
        # tbx_sim_step / tbx_train_chain log the DEFAULT DifferentiableReward (rewards.py:35-85: SmoothL1 position / speed terms, a
        # cosine SmoothL1 rotation term, no collision approximation); anything else would yield a plausible but wrong RolloutBuffer
        rc = to_attr(dict(differentiable_reward))
        if not rc.get("is_enabled", True) or rc.get("w_collision", 0) > 0 or not rc.get("use_il_loss", True):
            raise NotImplementedError("differentiable_reward: only the default configuration (is_enabled, use_il_loss, w_collision = 0) "
                                      "is implemented in the HIP state machine")
        for term, ang in (("l_pos", None), ("l_rot", "cosine"), ("l_spd", None)):
            t = rc[term]
            if t.get("criterion", "SmoothL1Loss") != "SmoothL1Loss" or (ang is not None and t.get("angular_type", ang) != ang):
                raise NotImplementedError(f"differentiable_reward.{term}: only SmoothL1Loss" + (" with angular_type=cosine" if ang else "") + " is implemented")
        pp = [(k, SceneCentricPreProcessing(time_step_current=time_step_current, data_size=data_size, **_strip_target(v)))
              for k, v in pre_processing.items()]
        kwargs = {"time_step_gt": time_step_gt}
        for _, m in pp:
            kwargs.update(m.model_kwargs)
        self.pre_processing = nn.Sequential(OrderedDict(pp))
        self.diffbar_reward = DifferentiableReward(**_strip_target(differentiable_reward))  # waymo_motion.py:71
        self.dynamics = Dynamics(navi_mode=kwargs["navi_mode"], **dynamics)
        mcfg = to_attr({**_strip_target(model), **kwargs, "action_dim": self.dynamics.action_dim})
        self.model = TrafficBots(**mcfg)
        self.teacher_forcing_training = TeacherForcing(**teacher_forcing_training)
        self.teacher_forcing_reactive_replay = TeacherForcing(**teacher_forcing_reactive_replay)
        self.teacher_forcing_joint_future_pred = TeacherForcing(**teacher_forcing_joint_future_pred)
        self._engine: Optional[RolloutEngine] = None
        self._engines: "OrderedDict[tuple, RolloutEngine]" = OrderedDict()  # engines by (shapes, schedule, weights version): begin_rollout
        self.engine_cache = 2  # engines kept (least recently used dropped); 0: a fresh engine + graph capture per rollout
        # which launches this module's hot path runs as (engine.Schedule): per module, not per process - e.g. a module with
        # bfloat16 K/V tables beside an fp32 one. Callers may assign a new one before encode_scene / rollout.
        self.schedule: Optional[engine.Schedule] = None

    @property
    def hp(self):
        return self.hparams

    # ------------------------------------------------------------------ once per scene
    def encode_scene(self, batch: Dict[str, Tensor], tl_valid_key: str = "sc/tl_valid", n_rollout: int = 1):
        """Map tokens + static traffic-light tokens (waymo_motion.py:316-323, 528-535). With n_rollout > 1 the
        traffic-light tokens are expanded per rollout while the map tokens stay shared (mp_batch_div)."""
        with engine.use(self.schedule):
            mp = self.model.mp_encoder(batch["sc/mp_valid"], batch["sc/mp_attr"], batch["sc/mp_pose"], batch["ref/mp_type"])
            r = lambda t: t.repeat_interleave(n_rollout, 0) if n_rollout > 1 else t
            tl = self.model.tl_encoder.pre_compute(tl_valid=r(batch[tl_valid_key]), tl_attr=r(batch["sc/tl_attr"]),
                                                   tl_pose=r(batch["sc/tl_pose"]), mp_batch_div=n_rollout, **mp)
        return mp, tl

    # ------------------------------------------------------------------ rollout
    def begin_rollout(self, ag_tokens: Dict[str, Tensor], mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor],
                      tl_state_gt: Tensor, teacher_forcing: TeacherForcing, rule_checker: TrafficRuleChecker, step_end: int,
                      stepwise: bool = False) -> RolloutEngine:
        """Step 0 of `rollout` (waymo_motion.py:218-231: teacher_forcing.init, dynamics.init, model.init): the device-resident
        simulation state. stepwise: driven through `forward` instead of the engine's own loop."""
        teacher_forcing.init(ag_valid=ag_tokens["gt_valid"], ag_pose=ag_tokens["gt_pose"], ag_motion=ag_tokens["gt_motion"],
                             tl_state=tl_state_gt, current_epoch=self.current_epoch)
        dev = ag_tokens["gt_pose"].device
        rc = self.hparams.differentiable_reward
        kw = dict(gt_valid=ag_tokens["gt_valid"], gt_pose=ag_tokens["gt_pose"], gt_motion=ag_tokens["gt_motion"],
                  tl_state_gt=tl_state_gt, tf_mask=teacher_forcing.ag_teacher_forcing, ag_type=ag_tokens["ag_type"],
                  ag_attr=ag_tokens["ag_attr"], ag_latent=ag_tokens["ag_latent"], ag_latent_valid=ag_tokens["ag_latent_valid"],
                  ag_navi=ag_tokens["ag_navi"], ag_navi_valid=ag_tokens["ag_navi_valid"], mp_tokens=mp_tokens,
                  tl_tokens=tl_tokens, map_valid=rule_checker.mp_valid, map_type=rule_checker.mp_type,
                  map_pos=rule_checker.mp_pos, map_dir=rule_checker.mp_dir, map_boundary=rule_checker.mp_boundary,
                  n_step=step_end, reward_weights=(rc.l_pos.weight, rc.l_rot.weight, rc.l_spd.weight),
                  ag_navi_log_prob=ag_tokens.get("ag_navi_log_prob"), stepwise=stepwise)
        eng = self._engine_for(kw, dev)
        self.dynamics.bind(eng)
        return eng

    def _engine_for(self, kw: dict, dev) -> RolloutEngine:
        """One engine per (shapes, schedule, weights), refilled in place: a loop over scenes (validation_step, waymo_motion.py:526)
        captures its hipGraphs once. `engine_cache = 0` turns the cache off (a fresh engine per rollout). kw: RolloutEngine.reset's."""
        mp_tokens, tl_tokens = kw["mp_tokens"], kw["tl_tokens"]
        sched = self.schedule if self.schedule is not None else engine.current()
        # (weights: every parameter's storage AND version - an optimizer step, load_state_dict, .to() or `p.data = ` all change one)
        key = (RolloutEngine.shape_key(**kw), dataclasses.astuple(sched), str(dev), self.training,
               hash(tuple((p.data_ptr(), p._version) for p in self.model.parameters())))
        eng = self._engines.get(key) if self.engine_cache > 0 else None
        if eng is not None:
            eng.refill(**kw)
            self._engines.move_to_end(key)
        else:
            eng = RolloutEngine(self.model, self.dynamics, dev, schedule=sched)
            if self.engine_cache > 0:
                # a cached engine is refilled in place with later scenes: it owns COPIES of the first scene's token dicts (one copy
                # per engine, not per rollout), so a caller still holding encode_scene()'s output never sees it change
                own = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items() if not k.startswith("_")}
                kw["mp_tokens"], kw["tl_tokens"] = own(mp_tokens), own(tl_tokens)
            eng.reset(**kw)
            if self.engine_cache > 0:
                eng.reused = True
                self._engines[key] = eng
                while len(self._engines) > self.engine_cache:
                    self._engines.popitem(last=False)
        self._engine = eng
        return eng

    @torch.no_grad()
    def rollout(self, ag_tokens: Dict[str, Tensor], mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor],
                tl_state_gt: Tensor, teacher_forcing: TeacherForcing, rule_checker: TrafficRuleChecker, step_end: int,
                deterministic_action: bool, player_policy=None, use_graph: bool = True, stepwise: bool = False) -> RolloutBuffer:
        """Reference signature (waymo_motion.py:206-217). Inference only: deterministic actions, no autograd.
        Default: the engine's device-side loop (one hipGraph replay per step). With a `player_policy` - a callable
        `ag_pose [n_sc, n_ag, 3] -> None | {"valid": [n_sc, n_ag] bool, "action": [n_sc, n_ag, 2]}` asked before every step,
        the hook the reference leaves as a todo (:240-241) - or stepwise=True, the reference's Python loop itself, one `forward`
        per step; both produce the same buffer (tested bit for bit)."""
        if not deterministic_action:
            raise NotImplementedError("stochastic actions are not used by any default entry point")
        stepwise = stepwise or player_policy is not None
        if not stepwise:
            eng = self.begin_rollout(ag_tokens, mp_tokens, tl_tokens, tl_state_gt, teacher_forcing, rule_checker, step_end)
            eng.run(step_end, use_graph=use_graph)
            return eng.buffer(self.hp.time_step_current, rule_checker=rule_checker)
        # ---- the reference's loop, statement by statement (waymo_motion.py:218-311)
        teacher_forcing.init(ag_valid=ag_tokens["gt_valid"], ag_pose=ag_tokens["gt_pose"], ag_motion=ag_tokens["gt_motion"],
                             tl_state=tl_state_gt, current_epoch=self.current_epoch)
        self.dynamics.init(tl_state=tl_state_gt, **ag_tokens)
        self.model.init()
        self._forward_steps = step_end  # (the log a forward-driven engine keeps; default hparams.time_step_end)
        dyn = self.dynamics
        buf = RolloutBuffer(step_end, self.hparams.time_step_current)
        lp0 = ag_tokens.get("ag_navi_log_prob")
        buf.add_navi_log_prob(torch.zeros_like(ag_tokens["gt_pose"][:, :, 0, 0]) if lp0 is None else lp0, ag_tokens["ag_navi_valid"])
        Tg, Tt = ag_tokens["gt_valid"].shape[-1], tl_state_gt.shape[2]
        for _step in range(1, step_end + 1):
            ag_override, tl_override = teacher_forcing.get(_step, dyn.ag_valid, dyn.ag_pose, dyn.ag_motion)
            player_override = player_policy(dyn.ag_pose) if player_policy is not None else None
            pred_dict, vis_dict = self.forward(mp_tokens, tl_tokens, ag_override, tl_override, player_override, deterministic_action)
            violation = rule_checker.check(pred_dict["pred_valid"], pred_dict["pred_pose"], pred_dict["pred_motion"], dyn.tl_state)
            _gt_valid = ag_tokens["gt_valid"][:, :, _step] if _step < Tg else None
            # reward and light NLL as tbx_sim_step logged them for this step (the expressions of self.diffbar_reward.get and of
            # -pred_tl_state_dist.log_prob: the engine's own loop hands out the same log)
            S, slot = self._engine.S, _step - 1
            r = S["out_reward"][:, :, slot]
            reward = {"diffbar_reward_valid": S["out_reward_valid"][:, :, slot].bool(), "diffbar_reward": r[..., 3],
                      "r_imitation_pos": r[..., 0], "r_imitation_rot": r[..., 1], "r_imitation_spd": r[..., 2],
                      "r_traffic_rule_approx": torch.zeros_like(r[..., 0])}
            nll = S["out_tl_nll"][:, :, slot]
            nll_invalid = torch.ones_like(tl_tokens["tl_token_invalid"]) if _step >= Tt else tl_tokens["tl_token_invalid"]
            buf.add(violation=violation, diffbar_reward=reward, tl_state_nll=nll, tl_state_nll_invalid=nll_invalid,
                    vis_dict=vis_dict, ag_override=ag_override, **pred_dict)
            dyn.disable_ag(violation, _gt_valid)
            dyn.disable_navi(violation)
        buf.finish()
        return buf

    def forward(self, mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor], ag_override: Dict[str, Tensor],
                tl_override: Dict[str, Tensor], player_override: Optional[Dict[str, Tensor]] = None,
                deterministic_action: bool = True):
        """Reference signature and semantics (waymo_motion.py:118-204): ONE closed-loop step on the simulation state that the
        reference's prologue - `self.dynamics.init(tl_state=tl_state_gt, **ag_tokens)`, `self.model.init()` (:228-229) - or
        `begin_rollout(..., stepwise=True)` set up: TrafficBots policy on the sliding windows,
        Dynamics.update_ag (player_override: {"valid", "action"} replaces the policy's physical action), then
        Dynamics.override_ag(ag_override {"valid", "pose", "motion"}) and override_tl(tl_override {"valid", "state"}).
        Returns (pred_dict, vis_dict) with the reference's keys. Rule checks and disable_ag / disable_navi stay with the caller
        (`self.dynamics.disable_ag(violation, gt_valid)`), as in the reference's `rollout`."""
        if self.dynamics._pending is not None:
            eng = self._engine_from_init(mp_tokens, tl_tokens)
        else:
            eng = self._engine
        if eng is None or not eng.stepwise or self.dynamics._eng is not eng:
            raise RuntimeError("forward() steps the simulation state that the reference's rollout prologue sets up - teacher_forcing.init, "
                               "self.dynamics.init(tl_state=..., **ag_tokens), self.model.init() (waymo_motion.py:218-229) - or "
                               "begin_rollout(..., stepwise=True)")
        if not deterministic_action:
            raise NotImplementedError("stochastic actions are not used by any default entry point")
        if eng._n_forward >= eng.T:
            raise RuntimeError(f"forward(): the engine logs {eng.T} steps per rollout (hparams.time_step_end, or rollout's step_end); "
                               "call self.dynamics.init(...) again to start another rollout")
        if not eng.reused:  # (a cached engine owns copies of the token dicts it was built from - the same values)
            eng.mp_tokens, eng.tl_tokens = mp_tokens, tl_tokens
        with torch.no_grad():
            slot = eng.forward_step(ag_override, tl_override, player_override)
        S, n, L = eng.S, eng.n, eng.L
        pred_valid = S["out_valid"][:, :, slot].bool()
        pred_dict = {"action_log_prob": eng.action_log_prob(S["out_valid"][:, :, slot]), "pred_valid": pred_valid,
                     "pred_pose": S["out_pose"][:, :, slot], "pred_motion": S["out_motion"][:, :, slot],
                     "pred_tl_state_dist": torch.distributions.Categorical(logits=S["tl_logits"].view(n, L, -1).clone())}
        vis_dict = {}
        if not self.training:
            dyn = self.dynamics
            vis_dict = {"pred_valid": dyn.ag_valid, "pred_pose": dyn.ag_pose.clone(), "pred_motion": dyn.ag_motion.clone(),
                        "action": S["out_action"][:, :, slot], "ag_navi": dyn.ag_navi, "ag_navi_valid": dyn.ag_navi_valid,
                        "navi_reached": dyn.mask_navi_reached, "tl_state": dyn.tl_state}
        return pred_dict, vis_dict

    def _engine_from_init(self, mp_tokens: Dict[str, Tensor], tl_tokens: Dict[str, Tensor]) -> RolloutEngine:
        """The step-wise engine of a rollout that was started the reference's way (Dynamics.init + TrafficBots.init), built by the
        first `forward` - the call that brings the tokens. Its tbx_sim_step copies of the outside-map / destination checks get
        tables that never fire: the caller's `rule_checker.check` evaluates them and `dynamics.disable_ag / disable_navi` apply them."""
        p = self.dynamics._pending
        dev = p["gt_pose"].device
        rc = self.hparams.differentiable_reward
        kw = dict(tf_mask=torch.zeros_like(p["gt_valid"]), mp_tokens=mp_tokens, tl_tokens=tl_tokens, map_valid=None, map_type=None, map_pos=None,
                  map_dir=None, map_boundary=None, n_step=int(getattr(self, "_forward_steps", None) or self.hparams.time_step_end),
                  reward_weights=(rc.l_pos.weight, rc.l_rot.weight, rc.l_spd.weight), stepwise=True, **p)
        eng = self._engine_for(kw, dev)
        self.dynamics.bind(eng)
        self._forward_steps = None
        return eng

    def _rule_checker(self, batch, ag_dest, tl_tokens, n_rollout: int = 1):
        """waymo_motion.py:399-412,497-510. Agent tensors per rollout, map tensors per scene (shared by its rollouts)."""
        r = (lambda t: t.repeat_interleave(n_rollout, 0)) if n_rollout > 1 else (lambda t: t)
        return TrafficRuleChecker(mp_boundary=batch["map/boundary"], mp_valid=batch["map/valid"], mp_type=batch["map/type"],
                                  mp_pos=batch["map/pos"], mp_dir=batch["map/dir"], ag_type=r(batch["ref/ag_type"]),
                                  ag_size=r(batch["ref/ag_size"]), ag_goal=None, ag_dest=ag_dest,
                                  tl_valid=tl_tokens["tl_token_valid"], tl_pose=tl_tokens["tl_token_pose"],
                                  disable_check=self.training)

    @torch.no_grad()
    def reactive_replay(self, batch, mp_tokens, tl_tokens, ag_latent, ag_latent_valid, ag_navi, ag_navi_valid,
                        teacher_forcing: TeacherForcing, deterministic_action: bool, step_end: Optional[int] = None,
                        use_graph: bool = True) -> RolloutBuffer:
        """waymo_motion.py:387-437: scene reconstruction given a complete episode."""
        ag_tokens = {"ag_type": batch["ref/ag_type"], "ag_size": batch["ref/ag_size"], "ag_attr": batch["sc/ag_attr"],
                     "gt_valid": batch["gt/ag_valid"], "gt_pose": batch["gt/ag_pose"], "gt_motion": batch["gt/ag_motion"],
                     "ag_latent": ag_latent, "ag_latent_valid": ag_latent_valid, "ag_navi": ag_navi, "ag_navi_valid": ag_navi_valid}
        buf = self.rollout(ag_tokens, mp_tokens, tl_tokens, batch["gt/tl_state"], teacher_forcing,
                           self._rule_checker(batch, ag_navi, tl_tokens), step_end or self.hp.time_step_end,
                           deterministic_action, use_graph=use_graph)
        buf.flatten_joint_future(1)
        return buf

    @torch.no_grad()
    def joint_future_pred(self, batch, mp_tokens, tl_tokens, ag_latent_dist: Optional[MyDist], ag_navi_dist: Optional[MyDist],
                          teacher_forcing: TeacherForcing, n_joint_future: int, step_end: Optional[int] = None,
                          use_graph: bool = True) -> RolloutBuffer:
        """waymo_motion.py:439-524: K parallel rollouts per scene from the history only. mp_tokens / tl_tokens exactly as
        `model.mp_encoder(...)` and `model.tl_encoder.pre_compute(tl_valid=..., **mp_tokens)` return them (validation_step,
        :528-535): the per-rollout repeat of the reference (:458-462) happens here, on a COPY of the light tokens (the caller's dicts
        stay as they are - `reactive_replay` has read the same ones), and only for the lights: the K rollouts of a scene share one copy
        of the map tokens and K/V tables (mp_batch_div). tl_tokens already expanded by `encode_scene(n_rollout=K)` are taken as is."""
        K = n_joint_future
        if K > 1 and tl_tokens.get("mp_batch_div", 1) == 1:
            tl_tokens = self._tl_tokens_per_rollout(tl_tokens, K)
        elif tl_tokens.get("mp_batch_div", 1) != K:
            raise ValueError(f"tl_tokens were expanded for {tl_tokens.get('mp_batch_div', 1)} rollouts per scene, n_joint_future = {K}")
        r = lambda t: t.repeat_interleave(K, 0)
        ag_tokens = {"ag_type": r(batch["ref/ag_type"]), "ag_size": r(batch["ref/ag_size"]), "ag_attr": r(batch["sc/ag_attr"]),
                     "gt_valid": r(batch["sc/ag_valid"]), "gt_pose": r(batch["sc/ag_pose"]), "gt_motion": r(batch["sc/ag_motion"])}
        if self.hp.joint_future_pred_deterministic_k0:
            det = torch.zeros_like(ag_tokens["gt_valid"][:, :, 0])
            det[::K] = True
        else:
            det = False
        if ag_latent_dist is None or ag_navi_dist is None:
            raise NotImplementedError("joint_future_pred: the default configuration predicts a latent and a destination per agent")
        ag_latent_dist.repeat_interleave_(K, 0)  # (in place, as the reference: :474,488)
        ag_tokens["ag_latent"] = ag_latent_dist.sample(deterministic=det)
        ag_tokens["ag_latent_valid"] = ag_latent_dist.valid
        ag_latent_log_prob = ag_latent_dist.log_prob(ag_tokens["ag_latent"]).masked_fill(~ag_tokens["ag_latent_valid"], 0)
        ag_navi_dist.repeat_interleave_(K, 0)
        ag_tokens["ag_navi"] = ag_navi_dist.sample(det)
        ag_tokens["ag_navi_valid"] = ag_navi_dist.valid
        ag_tokens["ag_navi_log_prob"] = ag_navi_dist.log_prob(ag_tokens["ag_navi"]).masked_fill(~ag_tokens["ag_navi_valid"], 0)
        checker = self._rule_checker(batch, ag_tokens["ag_navi"], tl_tokens, n_rollout=K)
        buf = self.rollout(ag_tokens, mp_tokens, tl_tokens, r(batch["sc/tl_state"]), teacher_forcing, checker,
                           step_end or self.hp.time_step_end, True, use_graph=use_graph)
        buf.flatten_joint_future(K)
        buf.compute_log_prob(ag_latent_log_prob)
        return buf

    @staticmethod
    def _tl_tokens_per_rollout(tl_tokens: Dict[str, Tensor], K: int) -> Dict[str, Tensor]:
        """The light tokens of `pre_compute` (one entry per scene) as `encode_scene(n_rollout=K)` makes them: every per-light tensor
        repeated K times along the batch (waymo_motion.py:460-462), the map targets still indexed per scene (mp_batch_div = K: the
        K-nearest map indices are local to a scene's M polylines). A new dict - the caller's is not touched -, bit-identical to
        running pre_compute on the repeated inputs (every row's K-nearest search and relative poses depend on that row alone)."""
        n = tl_tokens["tl_token_pose"].shape[0]
        out = {}
        for key, v in tl_tokens.items():
            if key.startswith("_"):
                continue  # per-module caches (K/V tables of the map tokens): rebuilt for the new dict
            if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n and key != "mp_feat_flat":
                out[key] = v.repeat_interleave(K, 0)
            else:
                out[key] = v
        out["mp_batch_div"] = K
        return out

    # ------------------------------------------------------------------ training
    def training_step(self, batch: Dict[str, Tensor], batch_idx: int, noise: Optional[Tensor] = None,
                      use_prior: Optional[Tensor] = None):
        """waymo_motion.py:313-385. KNARPE attention forward/backward, KNN and embeddings are HIP kernels; projections
        are library GEMMs; see train_graph.py for what is (not yet) fused."""
        from .. import train_graph

        with engine.use(self.schedule):
            out = train_graph.training_step(self, batch, noise=noise, use_prior=use_prior)
        for k, v in out.items():
            self.log(f"training/{k}", v, on_step=True)
        self.last_metrics = out
        return out["loss"]

    def configure_optimizers(self):
        """waymo_motion.py:820-838: AdamW, separate lr group for navi_predictor, StepLR."""
        navi, rest = [], []
        for k, v in self.named_parameters():
            (navi if "navi_predictor" in k else rest).append(v)
        o = _strip_target(self.hp.optimizer)
        # torch's fused multi-tensor AdamW on the device (one launch per chunk of parameters instead of ~10 foreach passes over ~700 tensors:
        # 1.9 -> 0.5 ms behind every training step's replay, tools/train_host_timeline.py); TBX_FUSED_ADAMW=0: the foreach form
        if "fused" not in o and "foreach" not in o and os.environ.get("TBX_FUSED_ADAMW", "1") != "0" and all(p.is_cuda for p in rest + navi):
            o["fused"] = True
        opt = torch.optim.AdamW(rest, **o)
        if navi:
            opt.add_param_group({"params": navi, "lr": self.hp.lr_navi})
        sch = torch.optim.lr_scheduler.StepLR(opt, **_strip_target(self.hp.lr_scheduler))
        return [opt], [{"scheduler": sch, "monitor": "val/loss", "interval": "epoch"}]
