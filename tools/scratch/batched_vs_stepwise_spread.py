"""Spread of the gradient difference between the time-batched training step and step-by-step autograd (tests/test_hip_training.py::
test_time_batched_training_rollout_equals_step_by_step_autograd, no dropout, C1 shape) over scene seeds, for TBX_LN_FWD=0/1:
worst max|gb - gs| / max(max|gs|, 3e-5) over the parameters and the parameter it belongs to."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
TG = import_module("trafficbots_amd.train_graph")
dev = torch.device("cuda:0")
n_sc, sizes, knn, n_steps = 2, (8, 64, 8), 4, 30
for ln_fwd in (False, True):
    TG.LN_FWD = ln_fwd
    for seed in (1, 2, 3, 4, 5, 6):
        cfg = tb.config.default_model_cfg(n_tgt_knn=knn)
        cfg["tf_cfg"]["dropout_p"] = 0.0
        cfg["mp_encoder"]["pl_encoder"]["mlp_dropout_p"] = 0.0
        cfg["add_navi_latent"]["mlp_dropout_p"] = 0.0
        scfg = tb.config.default_sim_cfg(p_training_rollout_prior=0.0)
        scfg["teacher_forcing_training"]["prob_forcing_agent"] = 0.0
        scfg["pre_processing"]["scene_centric"]["dropout_p_history"] = -1.0
        scfg["time_step_end"] = n_steps
        wm = W.WaymoMotion(model=cfg, data_size=tb.synthetic.DATA_SIZE, **scfg)
        tb.utils.det_fill(wm.model, 0)
        with torch.no_grad():
            for k, p in wm.model.named_parameters():
                if k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k:
                    p.mul_(0.02)
        wm = wm.to(dev).train()
        wm.attn_dropout_seed = torch.tensor([4242], dtype=torch.int64, device=dev)
        wm.tl_encoder_ahead, wm.fused_train_chain = True, True
        batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(n_sc, *sizes, seed=seed).items()}
        noise = torch.randn(n_sc, sizes[0], wm.model.latent_encoder.out_dim, generator=torch.Generator().manual_seed(5)).to(dev)
        use_prior = torch.zeros((), dtype=torch.bool, device=dev)
        res = {}
        for mode in (True, False):
            wm.time_batched_training = mode
            wm.zero_grad(set_to_none=True)
            loss = wm.training_step({k: v.clone() for k, v in batch.items()}, 0, noise=noise, use_prior=use_prior)
            loss.backward()
            res[mode] = {k: p.grad.clone() for k, p in wm.model.named_parameters() if p.grad is not None}
        worst = sorted(((float((res[True][k] - g).abs().max()) / max(float(g.abs().max()), 3e-5), k) for k, g in res[False].items()), reverse=True)
        print(f"LN_FWD={int(ln_fwd)} seed {seed}: worst {worst[0][0]:.2e} {worst[0][1]}; 2nd {worst[1][0]:.2e}; median {worst[len(worst)//2][0]:.2e}", flush=True)
