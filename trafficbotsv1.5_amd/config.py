"""Default hyper-parameters of the hot path, as plain attribute-dicts.

Values restate `configs/model/sim_agent.yaml` of the reference (interpolations resolved) and the
data dimensions of `src/data_modules/data_h5_womd.py:95-134` / `src/data_modules/scene_centric.py:28-37`.
No hydra / omegaconf dependency: the reference passes `DictConfig`s, this build accepts any mapping
with attribute access (SURVEY.md §8b "Config objects").
"""
import copy
from typing import Any, Dict


class AttrDict(dict):
    """dict with attribute access, `**` splat and deepcopy: what the model code needs of a DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(x: Any) -> Any:
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(to_attr(v) for v in x)
    return x


# data dimensions (reference data_h5_womd.py:95-101)
DATA_DIMS = dict(n_mp_pl_node=20, mp_attr_dim=11, tl_state_dim=5, ag_motion_dim=3, ag_attr_dim=6, n_step=91, n_step_hist=11)


def default_model_cfg(hidden_dim: int = 128, n_tgt_knn: int = 32, **overrides) -> AttrDict:
    """kwargs of `TrafficBots(...)` (reference traffic_bots.py:18-46) for the default sim_agent.yaml model."""
    pose = dict(theta_xy=1e3, theta_cs=1e1)
    cfg: Dict[str, Any] = dict(
        hidden_dim=hidden_dim,
        pairwise_relative=True,
        temp_window_size=11,
        n_tgt_knn=n_tgt_knn,
        dist_limit=500,
        tf_cfg=dict(
            d_model=hidden_dim, n_head=4, k_feedforward=4, dropout_p=0.1, bias=True, activation="relu",
            out_layernorm=False, apply_q_rpe=False,
        ),
        pose_rpe=dict(mode="pe_xy_yaw", **pose),
        mp_encoder=dict(
            n_layer_tf=8,
            pose_emb=dict(mode="mpa_pl", **pose),
            input_encoder=dict(mode="cat", n_layer=3, mlp_dropout_p=0, mlp_use_layernorm=False),
            pl_encoder=dict(pooling_mode="max_valid", n_layer=3, mlp_dropout_p=0.1, mlp_use_layernorm=False, use_pointnet=True),
        ),
        tl_encoder=dict(
            temp_stack_input=False, tl_lane_detach_mp_feature=True, n_layer_tf=4, k_tgt_knn_tl2tl=0.75,
            k_tgt_knn_tl2mp=0.75, k_dist_limit=0.5,
            pose_emb=dict(mode="pe_xy_yaw", **pose),
            input_encoder=dict(mode="add", n_layer=3, mlp_dropout_p=0, mlp_use_layernorm=False),
        ),
        tl_state_predictor=dict(detach_tl_feature=True, n_layer=3, rnn_dropout_p=0.1),
        ag_encoder=dict(
            n_layer_tf=4, k_tgt_knn_ag2mp=2.0, k_tgt_knn_ag2tl=0.8, k_tgt_knn_ag2ag=0.8, k_dist_limit=1.0,
            rnn_latent_temp_pool_mode="max_valid",
            pose_emb=dict(mode="pe_xy_yaw", **pose),
            input_encoder=dict(mode="cat", n_layer=3, mlp_dropout_p=0, mlp_use_layernorm=False),
        ),
        latent_encoder=dict(
            latent_dim=16, temporal_down_sample_rate=5, share_post_prior_encoders=False,
            latent_post=dict(dist_type="diag_gaus", n_cat=8, log_std=0.0, mlp_use_layernorm=False, n_layer=3, branch_type=False),
            latent_prior=dict(dist_type="std_gaus", n_cat=8, log_std=0.0, mlp_use_layernorm=False, n_layer=3, branch_type=False),
        ),
        navi_encoder=dict(dest_detach_mp_feature=True),
        navi_predictor=dict(
            detach_input=True, rnn_res_add=True, n_layer_tf=3, n_layer_mlp=3, mlp_use_layernorm=True, k_tgt_knn=1.0,
            k_dist_limit=1000, goal_log_std=2.0,
        ),
        add_navi_latent=dict(mode="cat", res_add=True, n_layer=3, mlp_use_layernorm=False, mlp_dropout_p=0.1),
        action_head=dict(log_std=-2, n_layer=3, branch_type=True, mlp_use_layernorm=False),
        # from SceneCentricPreProcessing.model_kwargs (scene_centric.py:28-37) + waymo_motion.py:70,79
        tl_mode="lane", navi_mode="dest", navi_dim=None,
        n_mp_pl_node=DATA_DIMS["n_mp_pl_node"], mp_attr_dim=DATA_DIMS["mp_attr_dim"],
        tl_state_dim=DATA_DIMS["tl_state_dim"], ag_motion_dim=DATA_DIMS["ag_motion_dim"],
        ag_attr_dim=DATA_DIMS["ag_attr_dim"], time_step_gt=90, action_dim=2,
    )
    cfg = to_attr(cfg)
    for k, v in overrides.items():
        cfg[k] = to_attr(v)
    return cfg


def default_sim_cfg(**overrides) -> AttrDict:
    """kwargs of `WaymoMotion(...)` besides `model` / `data_size` (sim_agent.yaml top level)."""
    cfg = dict(
        time_step_current=10, time_step_gt=90, time_step_end=90, time_step_sim_start=1, hidden_dim=128,
        p_training_rollout_prior=0.1, training_detach_model_input=True, training_deterministic_action=True,
        pred_navi_after_reached=False, n_joint_future_wosac=32, joint_future_pred_deterministic_k0=False,
        pre_processing=dict(scene_centric=dict(tl_mode="lane", navi_mode="dest", dropout_p_history=0.1)),
        teacher_forcing_training=dict(
            step_spawn_agent=10, step_warm_start=10, step_horizon=0, step_horizon_decrease_per_epoch=0,
            prob_forcing_agent=0.3, prob_forcing_agent_decrease_per_epoch=0.1, prob_scheduled_sampling=0,
            prob_scheduled_sampling_decrease_per_epoch=0, gt_sdc=False, threshold_xy=-1, threshold_yaw=-1, threshold_spd=-1,
        ),
        teacher_forcing_reactive_replay=dict(step_spawn_agent=90, step_warm_start=10),
        teacher_forcing_joint_future_pred=dict(step_spawn_agent=10, step_warm_start=10),
        dynamics=dict(
            use_veh_dynamics_for_all=False,
            veh=dict(max_acc=5, max_yaw_rate=1.5), cyc=dict(max_acc=6, max_yaw_rate=3), ped=dict(max_acc=7, max_yaw_rate=7),
        ),
        differentiable_reward=dict(
            w_collision=0, reduce_collsion_with_max=True, use_il_loss=True,
            l_pos=dict(weight=1e-1, criterion="SmoothL1Loss"),
            l_rot=dict(weight=1e1, criterion="SmoothL1Loss", angular_type="cosine"),
            l_spd=dict(weight=1e-1, criterion="SmoothL1Loss"),
        ),
        training_metrics=dict(
            w_vae_kl=1.0, kl_balance_scale=0.2, kl_free_nats=1.0, kl_for_unseen_agent=True, w_diffbar_reward=1.0,
            w_navi=1.0, w_tl_state=1.0, w_relevant_agent=0, p_loss_for_irrelevant=1.0, step_training_start=10,
            temporal_discount=-1, loss_for_teacher_forcing=True,
        ),
        optimizer=dict(lr=2e-4, weight_decay=1e-1, betas=(0.9, 0.95)),
        lr_navi=2e-4,
        lr_scheduler=dict(gamma=0.5, step_size=7),
    )
    cfg = to_attr(cfg)
    for k, v in overrides.items():
        cfg[k] = to_attr(v)
    return cfg
