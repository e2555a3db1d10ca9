"""`TrafficBots` (models/traffic_bots.py:17-221): model assembly with the reference's constructor, sub-module names
and state-dict keys. `forward` is the per-step policy: sliding windows -> tl encoder -> agent encoder -> navi/latent
fusion -> action head + tl-state head, every stage a HIP kernel launch on the current stream."""
from copy import deepcopy
from typing import Dict, Optional, Tuple

import os

import torch
from torch import Tensor, nn
from torch.distributions import Categorical, Independent, Normal

from .. import hip
from ..engine import emit_kv_tables
from .. import engine
from ..hip import BUF0, BUF1, Chain
from ..utils.pose_emb import PoseEmb
from .agent_encoder import AgentEncoder
from .latent_encoder import LatentEncoder
from .map_encoder import MapEncoder
from .modules.action_head import ActionHead
from .modules.add_navi_latent import AddNaviLatent
from .navigation import NaviEncoder, NaviPredictor
from .traffic_light import TrafficLightEncoder, TrafficLightStatePredictor


class TrafficBots(nn.Module):
    def __init__(self, hidden_dim: int, mp_attr_dim: int, tl_state_dim: int, ag_attr_dim: int, ag_motion_dim: int,
                 navi_mode: str, navi_dim: int, mp_encoder, tl_encoder, tl_state_predictor, ag_encoder, navi_encoder,
                 navi_predictor, latent_encoder, tf_cfg, time_step_gt: int, n_mp_pl_node: int, add_navi_latent, pose_rpe,
                 pairwise_relative: bool, temp_window_size: int, n_tgt_knn: int, dist_limit: float, tl_mode: str,
                 action_dim: int, action_head) -> None:
        super().__init__()
        if hidden_dim != 128 or tf_cfg["n_head"] != 4:
            raise NotImplementedError("the gfx950 kernels are built for hidden_dim=128, n_head=4")
        tl_encoder, ag_encoder = deepcopy(tl_encoder), deepcopy(ag_encoder)
        self.temp_window_size, self.hidden_dim, self.tl_state_dim = temp_window_size, hidden_dim, tl_state_dim
        self.pose_rpe = PoseEmb(pe_dim=hidden_dim, **pose_rpe) if pairwise_relative else None
        self.mp_encoder = MapEncoder(hidden_dim=hidden_dim, attr_dim=mp_attr_dim, n_mp_pl_node=n_mp_pl_node,
                                     pairwise_relative=pairwise_relative, n_tgt_knn=n_tgt_knn, dist_limit=dist_limit,
                                     tf_cfg=tf_cfg, pose_rpe=self.pose_rpe, **mp_encoder)
        tl_encoder.update(hidden_dim=hidden_dim, tl_state_dim=tl_state_dim, tl_mode=tl_mode, pairwise_relative=pairwise_relative,
                          n_tgt_knn=n_tgt_knn, dist_limit=dist_limit, tf_cfg=tf_cfg, temp_window_size=temp_window_size,
                          temp_encoder=mp_encoder["pl_encoder"])
        self.tl_encoder = TrafficLightEncoder(pose_rpe=self.pose_rpe, **tl_encoder)
        self.tl_state_predictor = TrafficLightStatePredictor(hidden_dim=hidden_dim, tl_state_dim=tl_state_dim,
                                                             temp_window_size=temp_window_size, **tl_state_predictor)
        ag_encoder.update(hidden_dim=hidden_dim, ag_attr_dim=ag_attr_dim, ag_motion_dim=ag_motion_dim,
                          pairwise_relative=pairwise_relative, n_tgt_knn=n_tgt_knn, dist_limit=dist_limit, tf_cfg=tf_cfg,
                          temp_window_size=temp_window_size, temp_encoder=mp_encoder["pl_encoder"])
        self.ag_encoder = AgentEncoder(pose_rpe=self.pose_rpe, **ag_encoder)
        self.latent_encoder = LatentEncoder(tl_encoder=tl_encoder, ag_encoder=ag_encoder, pose_rpe=self.pose_rpe,
                                            time_step_gt=time_step_gt, **latent_encoder)
        self.navi_encoder = NaviEncoder(hidden_dim=hidden_dim, navi_mode=navi_mode, navi_dim=navi_dim,
                                        pairwise_relative=pairwise_relative, mp_pose_emb=self.mp_encoder.pose_emb,
                                        pose_rpe=self.pose_rpe, **navi_encoder)
        self.navi_predictor = NaviPredictor(navi_mode=navi_mode, navi_dim=navi_dim, ag_encoder=ag_encoder,
                                            pose_rpe=self.pose_rpe, **navi_predictor)
        self.add_navi = AddNaviLatent(hidden_dim=hidden_dim, in_dim=hidden_dim, dummy=self.navi_encoder.dummy, **add_navi_latent)
        self.add_latent = AddNaviLatent(hidden_dim=hidden_dim, in_dim=self.latent_encoder.out_dim,
                                        dummy=self.latent_encoder.dummy, **add_navi_latent)
        self.action_head = ActionHead(hidden_dim=hidden_dim, action_dim=action_dim, **action_head)

    # ------------------------------------------------------------------ fused per-step policy on raw windows
    def policy_step(self, hist_valid: Tensor, hist_pose: Tensor, hist_motion: Tensor, hist_tl: Tensor, ag_attr6: Tensor,
                    ag_type_idx: Tensor, ag_latent: Tensor, latent_invalid: Tensor, dest: Tensor, navi_valid_u8: Tensor,
                    tl_tokens: Dict[str, Tensor], mp_tokens: Dict[str, Tensor], out: Dict[str, Tensor],
                    rollout_consts: Optional[Dict[str, Tensor]] = None) -> None:
        """One policy evaluation for all agents / lights. Inputs are the device-resident sliding windows (oldest first);
        writes out['action_mean'] [n*A,2], out['tl_logits'] [n*L,5] (+ out['ag_feat'], out['tl_feat']).
        No host synchronisation: capturable in a hipGraph. traffic_bots.py:188-221."""
        tl_kv = self.tl_policy(hist_tl, tl_tokens, out)
        self.agent_policy(hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, ag_latent, latent_invalid, dest,
                          navi_valid_u8, tl_tokens, mp_tokens, tl_kv, out, rollout_consts=rollout_consts)

    def tl_policy(self, hist_tl: Tensor, tl_tokens: Dict[str, Tensor], out: Dict[str, Tensor], prepared=None, tail_sim=None) -> Tensor:
        """The traffic-light half (traffic_bots.py:188-199): tl tokens of the window -> next-state logits in
        out['tl_logits'] and the per-layer K/V tables the agents' tl cross-attention reads (returned, out['tl_kv']).
        Reads no agent state, so the rollout engine runs it one step ahead on its own stream.
        tail_sim = dict(state, parts, attr, row_invalid): the lights' NEXT tbx_sim_step (+ tbx_tl_prep of their new windows) in the tail of
        the last layer's launch, behind the logits (tbx_tl_tail_t.sim_state: the rollout engine's one-queue step)."""
        n, L, _ = hist_tl.shape
        d = self.hidden_dim
        tl_inv = tl_tokens["tl_token_invalid_u8"]
        tl_kv = out.get("tl_kv")
        if tl_kv is None:
            tl_kv = out["tl_kv"] = torch.empty(n * L, 2 * d * len(self.ag_encoder.tf_ag2agmptl.layers), dtype=engine.kv_dtype(),
                                               device=hist_tl.device)

        def tl_tail(ch: Chain):
            emit_kv_tables(ch, self.ag_encoder.tl_kv_layers(), tl_kv)
            self.tl_state_predictor.emit(ch, tl_inv, out["tl_logits"])

        def tl_tail_mf():
            """`tl_tail` as tbx_tl_tail_t fields (the last layer's launch on the matrix path runs it), or None."""
            layers = self.ag_encoder.tl_kv_layers()
            lins = [t[0] for t in self.tl_state_predictor.mlp.linear_layers()]
            acts = [t[2] for t in self.tl_state_predictor.mlp.linear_layers()]
            if (len(layers) != 4 or len(lins) != 3 or acts != [True, True, False] or self.tl_state_predictor.tl_state_dim > 16
                    or any(tuple(l.weight.shape) != (d, d) for l in lins[:2]) or lins[2].weight.shape[1] != d
                    or any(ln is not None for _, ln, _ in self.tl_state_predictor.mlp.linear_layers())):
                return None
            pw = lambda w, b: hip.packed_weight(w, b, mfma32=True)
            w3, b3 = hip.stacked_linear([lins[2]], pad_out_to=16)
            return dict(kv_images=[pw(at.in_proj_weight[d:], at.in_proj_bias[d:]) for _, at in layers],
                        norms=[(nm.weight, nm.bias, nm.eps) for nm, _ in layers], kv_out=tl_kv,
                        mlp_images=[pw(lins[0].weight, lins[0].bias), pw(lins[1].weight, lins[1].bias), pw(w3, b3)], tl_invalid=tl_inv,
                        logits_out=out["tl_logits"], clamp=(-3.0, 3.0), sim=tail_sim)

        out["tl_feat"] = self.tl_encoder.encode(hist_tl, tl_tokens, tail=tl_tail, prepared=prepared, tail_mf=tl_tail_mf)  # prepared: TlEncoder.encode
        return tl_kv

    NAVI_AHEAD = os.environ.get("TBX_NAVI_AHEAD", "1") != "0"

    def agent_policy(self, hist_valid: Tensor, hist_pose: Tensor, hist_motion: Tensor, ag_attr6: Tensor, ag_type_idx: Tensor,
                     ag_latent: Tensor, latent_invalid: Tensor, dest: Tensor, navi_valid_u8: Tensor,
                     tl_tokens: Dict[str, Tensor], mp_tokens: Dict[str, Tensor], tl_kv: Tensor, out: Dict[str, Tensor],
                     aux_stream=None, rollout_consts: Optional[Dict[str, Tensor]] = None, fused_tail: Optional[dict] = None,
                     prep_ready: bool = False) -> None:
        """The agent half (traffic_bots.py:200-221): agent tokens attending to agents / map / tl K/V tables `tl_kv`, then
        navi + latent + action head -> out['action_mean'].
        fused_tail = dict(sim_state, parts): the rollout engine's request to run the agents' tbx_sim_step_parts and the NEXT step's
        tbx_agent_prep in the tail of the launch that produces the actions (tbx_heads_tail_t.sim_state / next_prep); whether it
        happened is reported in out["prep"]["_tail_fused"]. prep_ready: this step's tbx_agent_prep already ran (that tail)."""
        n, A, W = hist_valid.shape
        d = self.hidden_dim
        dev = hist_pose.device
        with engine.live_limit(engine.current().live_max_agents):  # (the agents' block leaves the one-row-per-workgroup layers earlier)
            # rollouts of one scene share its map tokens (mp_batch_div) and, in the rollout engine, its lights (tl_batch_div: the
            # light tokens are then per scene and "ag_mp_batch_div" carries the agents' map sharing)
            div = tl_tokens.get("ag_mp_batch_div", tl_tokens.get("mp_batch_div", 1))
            tl_inv = tl_tokens["tl_token_invalid_u8"]
            rc = rollout_consts or {}
            mp_flat = mp_tokens["mp_token_feature"].reshape(-1, d)
            # inference with an auxiliary stream: the navigation embedding mlp_in(navi feature) (navigation.py:65-79 +
            # add_navi_latent.py:43-50) reads nothing the agent layers produce - it runs there, behind the K-nearest searches, instead
            # of as the first five stages of the heads chain (same stages, same values)
            tile = engine.tile_rows_ok(n * A)  # large launches: the heads as one tbx_heads_tile launch (needs the embedding made ahead)
            # (Schedule.navi_rider: on small launches it rides in the agents' first-projection launch instead, with or without that stream)
            navi_ahead = (aux_stream is not None or tile or engine.current().navi_rider) and engine.DROP_CTX is None and self.NAVI_AHEAD

            def aux_tail(prep):
                if prep.get("navi_emb") is None:
                    prep["navi_emb"] = torch.empty(n * A, d, dtype=torch.float32, device=dev)
                cn = engine.row_chain(n * A, 4 * d + 4, big=(32, 4 * d + 4, d + 4, d + 4))
                self.navi_encoder.emit(cn, mp_flat, prep["navi_row"], prep["navi_pe"], dest_feature=rc.get("dest_feature"))
                prep["_navi_premasked"] = self.add_navi.emit_embed_buf(cn, prep["navi_emb"], navi_valid_u8.reshape(-1), mask_is_valid=True)
                cn.run(n * A)

            def navi_rider(prep, pose3: bool = False):
                """The same four stages as `aux_tail` as tbx_layer_tile_t's rider (agent_encoder.encode), or None. pose3: stage 0 reads the
                pose embedding the rider builds itself from prep["navi_pose3"] instead of prep["navi_pe"]."""
                l_pe = self.navi_encoder.mlp_pe.linear_layers()
                l_in = self.add_navi.mlp_in.linear_layers()
                if (rc.get("dest_feature") is None or len(l_pe) != 1 or len(l_in) != 3 or any(ln is not None for _, ln, _ in l_pe + l_in)
                        or any(tuple(t[0].weight.shape) != (d, d) for t in l_pe + l_in) or not all(act for _, _, act in l_in) or l_pe[0][2]):
                    return None
                if prep.get("navi_emb") is None:
                    prep["navi_emb"] = torch.empty(n * A, d, dtype=torch.float32, device=dev)
                prep["_navi_premasked"] = True
                src = (dict(pose3=prep["navi_pose3"], freqs=(self.pose_rpe.pe_xy.freqs, self.pose_rpe.pe_yaw.freqs)) if pose3
                       else dict(inp=prep["navi_pe"]))
                if pose3 and self.pose_rpe.out_dim != d:
                    return None
                return dict(add=rc["dest_feature"], out=prep["navi_emb"], valid=navi_valid_u8.reshape(-1),
                            images=[hip.packed_weight(t[0].weight, t[0].bias, mfma32=True) for t in l_pe + l_in], **src)

            def heads_tail(prep, mfma32: bool = False):
                """The heads as tbx_heads_tail_t / tbx_heads_tile_t fields when everything they read is at hand in the form the fused
                launch takes it: navigation embedding from the auxiliary stream (masked by its producer), latent embedding of the
                rollout (masked once), 3-layer adders without layernorm, the action head's stacked branches. mfma32: the images of
                tbx_heads_tile (large launches) instead of the gemv images of tbx_knarpe_dec_layer's tail."""
                ah, an, al = self.action_head, self.add_navi, self.add_latent
                if not (navi_ahead and prep.get("_navi_premasked") and rc.get("latent_premasked") and rc.get("latent_embedded") is not None
                        and ah.fused_branches and ah.masked_sum_store and len(ah.mlp_mean) == 3 and ah.out_dim <= 16):
                    return None
                pw = lambda w, b, **kw: hip.packed_weight(w, b, **(dict(mfma32=True) if mfma32 else dict(gemv=True)), **kw)
                lins = [[t[0] for t in mlp.linear_layers()] for mlp in ah.mlp_mean]
                w1, b1 = hip.stacked_linear([l[0] for l in lins])
                w2, b2 = hip.stacked_linear([l[1] for l in lins])
                w3, b3 = hip.stacked_linear([l[2] for l in lins], pad_out_to=16)
                imgs = [pw(t[0].weight, t[0].bias) for t in an.mlp.linear_layers()] + [pw(t[0].weight, t[0].bias) for t in al.mlp.linear_layers()]
                imgs += [pw(w1, b1), pw(w2, b2, groups=3), pw(w3, b3, groups=3)]
                hd = dict(images=imgs, navi_emb=prep["navi_emb"], latent_emb=rc["latent_embedded"], navi_valid=navi_valid_u8.reshape(-1),
                          latent_invalid=latent_invalid.reshape(-1), type_mask=prep["type_mask"], action_out=out["action_mean"])
                if fused_tail is not None and mfma32 and engine.current().fused_tail:
                    fused["args"] = self.ag_encoder.prep_args(hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, prep, dest, mp_tokens, div)
                    hd.update(sim_state=fused_tail["sim_state"], sim_parts=fused_tail["parts"], next_prep=fused["args"])
                return hd

            fused = {}
            feat, prep = self.ag_encoder.encode(hist_valid, hist_pose, hist_motion, ag_attr6, mp_tokens, tl_inv,
                                                tl_tokens["tl_token_pose"], tl_kv, prep=out.get("prep"), ag_type_idx=ag_type_idx,
                                                dest=dest, mp_batch_div=div, tl_batch_div=tl_tokens.get("tl_batch_div", 1),
                                                aux_stream=aux_stream, navi_rpe=self.pose_rpe, aux_tail=aux_tail if navi_ahead else None,
                                                heads_tail=heads_tail if navi_ahead else None,
                                                navi_rider=navi_rider if navi_ahead else None, prep_ready=prep_ready)
            out["prep"], out["ag_feat"] = prep, feat
            prep["_tail_fused"] = bool(prep.get("_heads_done")) and "args" in fused
            prep["_tail_keep"] = fused.get("args")  # (the ctypes structure outlives the launch call anyway; kept for clarity)
            if prep.get("_heads_done"):  # the last layer's launch already ran the adders and the action head (engine.run_block)
                return
            if tile:
                hd = heads_tail(prep, mfma32=True)
                if hd is not None:
                    hip.heads_tile(feat, hd)
                    return
            if engine.DROP_CTX is not None and engine.tile_rows_ok(n * A, keyed_dropout=True) and rc.get("dest_feature") is not None:
                hd = self._heads_tile_raw(prep, rc, ag_latent, latent_invalid, navi_valid_u8, out, n * A)
                if hd is not None:  # training's stepping pass: embeddings, adders (with their keyed dropouts) and action head in ONE launch
                    hip.heads_tile(feat, hd)
                    return
            navi_pe = prep["navi_pe"]
            ch = engine.row_chain(n * A, 4 * d + 4, big=(32, 4 * d + 4, d + 4, d + 4))
            ch.load(feat, BUF1, 0, n=d)
            if navi_ahead:  # (its stream was joined before the first attention launch - run_block - with this chain already enqueued on it)
                self.add_navi.emit(ch, navi_valid_u8.reshape(-1), mask_is_valid=True, z_embedded=prep["navi_emb"],
                                   z_premasked=bool(prep.get("_navi_premasked")))
            else:
                self.navi_encoder.emit(ch, mp_flat, prep["navi_row"], navi_pe, dest_feature=rc.get("dest_feature"))
                self.add_navi.emit(ch, navi_valid_u8.reshape(-1), mask_is_valid=True)
            if engine.DROP_CTX is not None:  # training's stepping pass: 12 DROPOUT stages more than a program holds - two launches
                mid = torch.empty_like(feat)
                ch.store(BUF1, 0, d, mid)
                ch.run(n * A)
                ch = Chain(32, 4 * d + 4, d + 4, d + 4) if n * A >= 16384 else Chain(16, 4 * d + 4)
                ch.load(mid, BUF1, 0, n=d)
            self.add_latent.emit(ch, latent_invalid, ag_latent, z_embedded=rc.get("latent_embedded"), z_premasked=bool(rc.get("latent_premasked")))
            self.action_head.emit(ch, prep["type_mask"], out["action_mean"])
            ch.run(n * A)

    def _heads_tile_raw(self, prep, rc, ag_latent, latent_invalid, navi_valid_u8, out, rows: int):
        """tbx_heads_tile_t with raw = 1 (training's stepping pass at >= Schedule.tile_min_rows rows): the heads' two row chains -
        navigation / latent embeddings, the adders with their 12 keyed dropouts, the action head - as one launch; None where the
        modules are not of the shape the kernel is built for. The dropout site ids are taken in the chains' order."""
        ah, an, al, d = self.action_head, self.add_navi, self.add_latent, self.hidden_dim
        l_pe = self.navi_encoder.mlp_pe.linear_layers()
        lin = lambda mlp: [t[0] for t in mlp.linear_layers()]
        ok_mlp = lambda mlp, k0: (len(lin(mlp)) == 3 and all(t[1] is None and t[2] for t in mlp.linear_layers())
                                  and [tuple(l.weight.shape) for l in lin(mlp)] == [(d, k0), (d, d), (d, d)])
        if not (len(l_pe) == 1 and l_pe[0][1] is None and not l_pe[0][2] and tuple(l_pe[0][0].weight.shape) == (d, d) and d == 128
                and ok_mlp(an.mlp_in, d) and ok_mlp(an.mlp, 2 * d) and ok_mlp(al.mlp, 2 * d) and ok_mlp(al.mlp_in, al.in_dim) and al.in_dim <= 16
                and ah.fused_branches and ah.masked_sum_store and len(ah.mlp_mean) == 3 and ah.out_dim <= 16 and prep.get("navi_pe") is not None):
            return None
        ps = [an.mlp_in.dropout_p] * 3 + [an.mlp.dropout_p] * 3 + [al.mlp_in.dropout_p] * 3 + [al.mlp.dropout_p] * 3
        sites = [engine.drop_site(p_) for p_ in ps]  # (the order add_navi.emit / add_latent.emit take them in)
        live = [s_ for s_ in sites if s_ is not None]
        if live and len({s_[0] for s_ in live}) != 1:
            raise NotImplementedError("tbx_heads_tile: one dropout probability for the adders' MLPs")
        drop = dict(p=live[0][0], seed=live[0][1], step=live[0][3], sites=[None if s_ is None else s_[2] for s_ in sites]) if live else None
        pw = lambda w, b, **kw: hip.packed_weight(w, b, mfma32=True, **kw)
        lins = [lin(mlp) for mlp in ah.mlp_mean]
        w1, b1 = hip.stacked_linear([l[0] for l in lins])
        w2, b2 = hip.stacked_linear([l[1] for l in lins])
        w3, b3 = hip.stacked_linear([l[2] for l in lins], pad_out_to=16)
        imgs = [pw(l.weight, l.bias) for l in lin(an.mlp)] + [pw(l.weight, l.bias) for l in lin(al.mlp)]
        imgs += [pw(w1, b1), pw(w2, b2, groups=3), pw(w3, b3, groups=3)]
        li = lin(al.mlp_in)
        raw_imgs = ([pw(l_pe[0][0].weight, l_pe[0][0].bias)] + [pw(l.weight, l.bias) for l in lin(an.mlp_in)]
                    + [pw(hip.padded_weight(li[0].weight, 32), li[0].bias), pw(li[1].weight, li[1].bias), pw(li[2].weight, li[2].bias)])
        z = ag_latent.reshape(rows, -1)
        if z.shape[1] < 16 or z.stride(0) % 4 or z.data_ptr() % 16:
            z = torch.nn.functional.pad(z, (0, 16 - z.shape[1])) if z.shape[1] < 16 else z.contiguous()
        return dict(images=imgs, navi_valid=navi_valid_u8.reshape(-1), latent_invalid=latent_invalid.reshape(-1), type_mask=prep["type_mask"],
                    action_out=out["action_mean"],
                    raw=dict(navi_pe=prep["navi_pe"], dest_feature=rc["dest_feature"], latent_z=z, images=raw_imgs, drop=drop))

    @torch.no_grad()
    def rollout_constants(self, ag_latent: Tensor, dest: Tensor, mp_tokens: Dict[str, Tensor], mp_batch_div: int = 1,
                          latent_invalid: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """What the heads chain would recompute identically at every step of a rollout: mlp_in(latent) of `add_latent` and
        mlp_mp(map feature of the destination) of the navi encoder (latent and destination are fixed per rollout:
        waymo_motion.py:232-311 with pred_navi_after_reached off). Same kernels, same values - evaluated once."""
        n, A = dest.shape
        d, dev, M = self.hidden_dim, ag_latent.device, mp_tokens["mp_token_pose"].shape[1]
        lat = torch.empty(n * A, d, dtype=torch.float32, device=dev)
        ch = Chain(16, 4 * d + 4)
        premasked = self.add_latent.emit_embed(ch, ag_latent.reshape(n * A, -1).float().contiguous(), lat,
                                               None if latent_invalid is None else latent_invalid.reshape(-1))
        ch.run(n * A)
        rows = ((torch.arange(n, device=dev) // mp_batch_div).unsqueeze(1) * M + dest).reshape(-1).to(torch.int32).contiguous()
        dst = torch.empty(n * A, d, dtype=torch.float32, device=dev)
        ch = Chain(16, 4 * d + 4)
        self.navi_encoder.emit_dest_feature(ch, mp_tokens["mp_token_feature"].reshape(-1, d), rows, dst)
        ch.run(n * A)
        return {"latent_embedded": lat, "dest_feature": dst, "latent_premasked": premasked}

    # ------------------------------------------------------------------ reference per-step API
    def init(self) -> None:
        self.hist_ag_valid = self.hist_ag_pose = self.hist_ag_motion = self.hist_tl_state = None
        self.navi_feature = None
        self.tl_state_predictor.init()

    def _append_hist(self, ag_valid, ag_pose, ag_motion, tl_state) -> None:
        new = [ag_valid.unsqueeze(2), ag_pose.unsqueeze(2), ag_motion.unsqueeze(2), tl_state.unsqueeze(2)]
        if self.hist_ag_valid is None:
            self.hist_ag_valid, self.hist_ag_pose, self.hist_ag_motion, self.hist_tl_state = new
        else:
            W = self.temp_window_size
            self.hist_ag_valid = torch.cat([self.hist_ag_valid, new[0]], 2)[:, :, -W:]
            self.hist_ag_pose = torch.cat([self.hist_ag_pose, new[1]], 2)[:, :, -W:]
            self.hist_ag_motion = torch.cat([self.hist_ag_motion, new[2]], 2)[:, :, -W:]
            self.hist_tl_state = torch.cat([self.hist_tl_state, new[3]], 2)[:, :, -W:]

    def forward(self, ag_valid: Tensor, ag_pose: Tensor, ag_motion: Tensor, ag_attr: Tensor, ag_type: Tensor,
                ag_latent: Optional[Tensor], ag_latent_valid: Optional[Tensor], ag_navi: Optional[Tensor],
                ag_navi_valid: Tensor, ag_navi_updated: bool, tl_state: Tensor, tl_tokens: Dict[str, Tensor],
                mp_tokens: Dict[str, Tensor]) -> Tuple[Independent, Categorical]:
        """Reference signature (traffic_bots.py:151-166)."""
        self._append_hist(ag_valid, ag_pose, ag_motion, tl_state)
        n, A = ag_valid.shape
        L, W, dev = tl_state.shape[1], self.temp_window_size, ag_pose.device
        hv, hp, hm = self.ag_encoder.pad_hist(self.hist_ag_valid, self.hist_ag_pose, self.hist_ag_motion, W)
        ht = self.tl_encoder.states_to_hist(self.hist_tl_state, W)
        out = dict(action_mean=torch.empty(n * A, 2, dtype=torch.float32, device=dev),
                   tl_logits=torch.empty(n * L, self.tl_state_dim, dtype=torch.float32, device=dev))
        lat_inv = (~ag_latent_valid).reshape(-1).to(torch.uint8).contiguous()
        type_idx = ag_type.to(torch.uint8).argmax(-1).to(torch.uint8).contiguous()
        self.policy_step(hv, hp, hm, ht, ag_attr.float().contiguous(), type_idx, ag_latent.reshape(n * A, -1).float().contiguous(),
                         lat_inv, ag_navi.contiguous(), ag_navi_valid.to(torch.uint8).contiguous(), tl_tokens, mp_tokens, out)
        mean = out["action_mean"].view(n, A, 2)
        log_std = self.action_head.masked_log_std(ag_valid, ag_type)
        return Independent(Normal(mean, log_std.exp()), 1), Categorical(logits=out["tl_logits"].view(n, L, -1))
