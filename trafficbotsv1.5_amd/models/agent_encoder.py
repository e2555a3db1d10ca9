"""`AgentEncoder` (models/agent_encoder.py:15-466), HPTR variant (`_forward_hptr`): per step, the agents' W-step
windows -> local-frame PointNet tokens -> 4 dec_cross_attn layers over [K nearest agents | K nearest map tokens ++
K nearest traffic lights]."""
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import hip
from ..engine import current as engine_current, drop_site, emit_mlp, D, FIRST_PROJ_LDW, SelfKnn, emit_first_proj, emit_pointnet, first_proj_buffers, front_ok, front_proj_buffers, kv_dtype, kv_tables, run_block, tile_proj_part, tile_rows_ok, tile_small_ok
from ..hip import BUF0, BUF1, Chain, Seg
from ..utils.pose_emb import PoseEmb
from .modules.input_encoder import InputEncoder
from .modules.polyline_encoder import PolylineEncoder
from .modules.transformer_rpe import TransformerBlockRPE


class AgentEncoder(nn.Module):
    def __init__(self, hidden_dim: int, ag_attr_dim: int, ag_motion_dim: int, pairwise_relative: bool, pose_emb,
                 input_encoder, temp_encoder, pose_rpe: PoseEmb, tf_cfg, n_tgt_knn: int, k_tgt_knn_ag2ag: float,
                 k_tgt_knn_ag2mp: float, k_tgt_knn_ag2tl: float, dist_limit: float, k_dist_limit: float, n_layer_tf: int,
                 temp_window_size: int, rnn_latent_temp_pool_mode: str) -> None:
        super().__init__()
        if not pairwise_relative or temp_window_size <= 0 or input_encoder["mode"] != "cat":
            raise NotImplementedError("the MI355X path implements the default pairwise-relative HPTR agent encoder")
        self.hidden_dim, self.temp_window_size = hidden_dim, temp_window_size
        self.n_tgt_knn_ag2ag = int(n_tgt_knn * k_tgt_knn_ag2ag)
        self.n_tgt_knn_ag2mp = int(n_tgt_knn * k_tgt_knn_ag2mp)
        self.n_tgt_knn_ag2tl = int(n_tgt_knn * k_tgt_knn_ag2tl)
        self.dist_limit = dist_limit * k_dist_limit
        self.pose_emb = PoseEmb(pe_dim=hidden_dim // 2, **pose_emb)
        self.pose_rpe = pose_rpe
        attr_dim = ag_attr_dim + ag_motion_dim + temp_window_size
        assert ag_attr_dim == 6 and ag_motion_dim == 3 and attr_dim <= 32
        self.register_buffer("hist_ohe", torch.eye(temp_window_size))
        self.temp_encoder = PolylineEncoder(hidden_dim=hidden_dim, tf_cfg=tf_cfg, **temp_encoder)
        self.tf_ag2agmptl = TransformerBlockRPE(n_layer=n_layer_tf, mode="dec_cross_attn", d_rpe=pose_rpe.out_dim, **tf_cfg)
        self.input_encoder = InputEncoder(hidden_dim=hidden_dim, attr_dim=attr_dim, pe_dim=self.pose_emb.out_dim, **input_encoder)

    # ---- static per scene
    def kv_mp(self, mp: Dict[str, Tensor], refresh: bool = False) -> Tensor:
        """K/V tables of the map tokens for this encoder's ag2mp layers; cached in the map-token dict (refresh: recomputed into the
        cached tensor - the token features were overwritten in place, RolloutEngine.refill)."""
        cache = mp.setdefault("_kv_ag", {})
        key = (id(self), kv_dtype())  # (an fp32 engine and a bf16-table engine may share one token dict: a table per element type)
        if key not in cache or refresh:
            feat = mp["mp_token_feature"].reshape(-1, self.hidden_dim).contiguous()
            cache[key] = kv_tables(feat, [(l.norm_tgt, l.attn) for l in self.tf_ag2agmptl.layers], out=cache.get(key))
        return cache[key]

    def tl_kv_layers(self):
        return [(l.norm_tgt, l.attn) for l in self.tf_ag2agmptl.layers]

    def alloc_prep(self, n: int, A: int, dev, with_heads: bool) -> Dict[str, Tensor]:
        W, pe = self.temp_window_size, self.pose_emb.out_dim
        f32, u8 = torch.float32, torch.uint8
        out = dict(tok_pose=torch.empty(n, A, 3, dtype=f32, device=dev), tok_invalid=torch.empty(n, A, dtype=u8, device=dev),
                   attr=torch.empty(n * A * W, 32, dtype=f32, device=dev), pe=torch.empty(n * A * W, pe, dtype=f32, device=dev),
                   row_invalid=torch.empty(n * A * W, dtype=u8, device=dev))
        if with_heads:
            out.update(type_mask=torch.empty(3, n * A, dtype=u8, device=dev), navi_pose3=torch.empty(n * A, 3, dtype=f32, device=dev),
                       navi_row=torch.empty(n * A, dtype=torch.int32, device=dev))
        return out

    def _prep_call(self, hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, prep, dest, mp, mp_batch_div):
        return ((hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, self.pose_emb.pe_xy.freqs, self.pose_emb.pe_yaw.freqs,
                 self.pose_emb.out_dim, prep),
                dict(dest=dest, mp_tok_pose=mp["mp_token_pose"] if dest is not None else None, n_mp=mp["mp_token_pose"].shape[1],
                     mp_batch_div=mp_batch_div))

    def run_prep(self, *a) -> None:
        """tbx_agent_prep of the windows into `prep` (agent_encoder.py:130-159's inputs + token poses, type masks, navi rows)."""
        args, kw = self._prep_call(*a)
        hip.agent_prep(*args, **kw)

    def prep_args(self, *a):
        """The same call as a tbx_agent_prep_args_t (the fused step tail of tbx_knarpe_dec_layer runs it for the next step)."""
        args, kw = self._prep_call(*a)
        return hip.agent_prep_args(*args, **kw)

    def encode(self, hist_valid: Tensor, hist_pose: Tensor, hist_motion: Tensor, ag_attr6: Tensor, mp: Dict[str, Tensor],
               tl_invalid_u8: Tensor, tl_pose: Tensor, tl_kv: Tensor, prep: Optional[Dict[str, Tensor]] = None,
               ag_type_idx: Optional[Tensor] = None, dest: Optional[Tensor] = None, mp_batch_div: int = 1, tl_batch_div: int = 1,
               tail: Optional[Callable[[Chain], None]] = None, aux_stream=None, navi_rpe=None,
               aux_tail: Optional[Callable[[Dict[str, Tensor]], None]] = None,
               heads_tail: Optional[Callable[[Dict[str, Tensor]], Optional[dict]]] = None,
               navi_rider: Optional[Callable[[Dict[str, Tensor]], Optional[dict]]] = None,
               prep_ready: bool = False) -> Tuple[Tensor, Dict[str, Tensor]]:
        """hist_* [n,A,W(,3)] oldest first (u8 / f32); tl_kv = K/V tables of this step's tl tokens [n*L, 4*256]
        ([n/tl_batch_div * L, ..] with tl_pose / tl_invalid_u8 [n/tl_batch_div, L, ..] when the rollouts of a scene share its lights).
        -> ag_token_feature [n*A, d] and the prep dict (token pose/invalid, type masks, navi rows).
        aux_stream: the three K-nearest searches (they need the token poses only) run there while this stream runs the
        temporal PointNet of the agents' windows; their outputs live in `prep` across steps. navi_rpe: the PoseEmb of the navigation
        encoder - the embedding of the destination's relative pose (prep["navi_pe"], an input of the heads chain that depends on
        agent_prep only) is then computed on that stream too instead of between the last layer and the heads. heads_tail(prep): the
        caller's heads as tbx_heads_tail_t fields for the last layer's launch (see engine.run_block). aux_tail(prep): more
        work of the caller's that needs nothing but `prep` (the navigation embedding of the heads), enqueued on that stream after it.
        navi_rider(prep) -> that same work as the rider of the first projection's tbx_layer_tile launch (small launches,
        Schedule.navi_rider): everything then runs on this stream - a cross-queue edge of the captured graph costs ~5 us of idle
        queue at its source and ~6-12 us at its destination (profiles/r03_c2_two_stream_timeline.txt)."""
        n, A, W = hist_valid.shape
        assert W == self.temp_window_size
        dev, d, rp = hist_pose.device, self.hidden_dim, self.pose_rpe
        M, L = mp["mp_token_pose"].shape[1], tl_pose.shape[1]
        if prep is None:
            assert not prep_ready
            prep = self.alloc_prep(n, A, dev, with_heads=dest is not None)
        if not prep_ready:  # (prep_ready: the previous step's last launch already ran this step's tbx_agent_prep - its fused tail)
            self.run_prep(hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, prep, dest, mp, mp_batch_div)
        tok_pose, tok_inv = prep["tok_pose"], prep["tok_invalid"]
        mp_inv = mp.get("mp_token_invalid_u8")
        if mp_inv is None:
            mp_inv = mp["mp_token_invalid_u8"] = mp["mp_token_invalid"].to(torch.uint8).contiguous()
        # Order of enqueueing = order of the captured graph's nodes = order the runtime launches them in: the window PointNet
        # (critical path: its pooled rows feed the first projection) goes in BEFORE the three K-nearest searches of the auxiliary
        # stream, which are only needed by the first attention call (measured on the two-stream timeline: the chain started 20 us
        # after agent_prep ended because the three searches were launched ahead of it).
        main = torch.cuda.current_stream()
        # Schedule.knn_main: the searches stay on this stream (a cross-queue dependency costs ~10 us of idle queue on each side of
        # it - measured on the two-stream timeline - more than the 15 us launch it hides); the auxiliary stream then only makes the
        # heads' navigation embedding and is joined before the last layer
        x = torch.empty(n * A, d, dtype=torch.float32, device=dev)
        fp = first_proj_buffers(n * A, dev, hip.group_tile_rows(W, n * A))  # small launches: layer 0's projections in the windows' launch too
        want_pe = navi_rpe is not None and dest is not None
        pe_rides = want_pe and engine_current().pe_rides and n * A < 4096  # the destination's pose embedding in the searches' launch
        if pe_rides and prep.get("navi_pe") is None:
            prep["navi_pe"] = torch.empty(n * A, navi_rpe.out_dim, dtype=torch.float32, device=dev)
        rider = None
        if (navi_rider is not None and engine_current().navi_rider and want_pe and aux_tail is not None and fp is None
                and (tile_small_ok() or tile_rows_ok(n * A, keyed_dropout=True))):
            # (large launches: the destination's pose embedding is built inside the rider - nothing of it rides on the searches)
            rider = navi_rider(prep, pose3=not pe_rides)  # (None: the modules are not of the shape the rider is built for)
        use_rider = rider is not None
        # Schedule.knn_aux_big: at large launches the three searches (35 us at 4096 rows) run on the auxiliary stream beside the window
        # PointNet + first projection (50 us) - there the two cross-queue edges cost less than the launches they hide
        knn_aux = aux_stream is not None and engine_current().knn_aux_big and n * A >= 4096
        if use_rider and not knn_aux:
            aux_stream = None  # nothing of this step runs beside this stream
        knn_main = aux_stream is not None and engine_current().knn_main and not knn_aux
        if aux_stream is not None and not knn_main:
            aux_stream.wait_stream(main)  # fork: the searches depend on agent_prep only
        ie = self.input_encoder
        wt_ = self._window_tile_images(prep["attr"].shape[1]) if (fp is None and W <= 16 and (tile_rows_ok(n * A, keyed_dropout=True) or tile_small_ok())) else None
        ch = Chain(hip.group_tile_rows(W, n * A), d + 4 if fp is None else FIRST_PROJ_LDW)
        # tbx_front: window PointNet + first projection (+ rider) + the searches as ONE launch - when nothing of this step runs beside
        # this stream (the navigation embedding rides, or is not wanted here) and no keyed dropout is asked for
        sites0 = [drop_site(m.dropout_p) for m in self.temp_encoder.mlp_layers] if wt_ is not None else []
        fused_front = (wt_ is not None and front_ok(n * A) and aux_stream is None and all(s_ is None for s_ in sites0)
                       and (use_rider or aux_tail is None))
        if fused_front:
            if use_rider and rider.get("pose3") is None:
                rider = navi_rider(prep, pose3=True)  # (the searches' pose-embedding job runs in the SAME launch: not an input of it)
                fused_front = rider is not None
        if fused_front:
            ch = None
        elif wt_ is not None:  # the whole temporal PointNet as one tbx_window_tile launch (training's stepping pass: with its keyed dropouts)
            sites = sites0  # the ids emit_pointnet's DROPOUT stages would take
            drop = None
            if any(s_ is not None for s_ in sites):
                assert all(s_ is not None for s_ in sites)
                drop = dict(p=sites[0][0], seed=sites[0][1], step=sites[0][3], sites=[s_[2] for s_ in sites])
            hip.window_tile(prep["attr"], prep["pe"], prep["row_invalid"], wt_[0], wt_[1], W, x, drop=drop)
            ch = None
        elif (ie.mode == "cat" and len(ie.mlp.linear_layers()) == 3 and ie.mlp.output_dim % 16 == 0 and prep["attr"].shape[1] % 4 == 0
                and ie.mlp.output_dim + ie.pe_dim <= d):
            # attribute rows (agent_prep zero-fills them to 32 columns) and pose embeddings in ONE load stage; the embedding
            # goes straight to where the concatenation wants it (BUF1[:, out:out+pe] - the three MLP stages ping-pong
            # BUF0 -> BUF1 -> BUF0 -> BUF1 and write whole 16-column tiles of [0, out) only)
            ch.load2(prep["attr"], BUF0, 0, prep["pe"], BUF1, ie.mlp.output_dim)
            cur = emit_mlp(ch, ie.mlp, BUF0, 0)
            assert cur == BUF1
        elif ch is not None:
            cur = ie.emit(ch, prep["attr"], prep["pe"])
        if ch is not None:
            kept = emit_pointnet(ch, self.temp_encoder, prep["row_invalid"], x, x_buf=cur, keep=fp is not None)
            if fp is not None:
                emit_first_proj(ch, self.tf_ag2agmptl, fp, kept)
            ch.run(n * A * W, group_rows=W)
        # the agents' KNN sets change every step: only the relative poses are produced (12 B per pair); the attention
        # kernel rebuilds the 128-d embedding in registers in each of the 4 layers
        kw = dict(want_rel_pose=True, want_emb=False)
        with torch.cuda.stream(aux_stream if aux_stream is not None and not knn_main else main):
            common = dict(src_pose=tok_pose, src_invalid=tok_inv, dist_limit=self.dist_limit, **kw)
            jobs = [dict(common, tgt_pose=mp["mp_token_pose"], tgt_invalid=mp_inv, k=self.n_tgt_knn_ag2mp, tgt_batch_div=mp_batch_div,
                         out=prep.get("_knn_am")),  # the longest search first
                    dict(common, tgt_pose=tok_pose, tgt_invalid=tok_inv, k=self.n_tgt_knn_ag2ag, out=prep.get("_knn_aa")),
                    dict(common, tgt_pose=tl_pose, tgt_invalid=tl_invalid_u8, k=self.n_tgt_knn_ag2tl, tgt_batch_div=tl_batch_div,
                         out=prep.get("_knn_at"))]
            # the destination's pose embedding as a job of the searches' launch (pe_rides) - unless the rider builds it itself
            pj = dict(pose3=prep["navi_pose3"], freqs_xy=navi_rpe.pe_xy.freqs, freqs_yaw=navi_rpe.pe_yaw.freqs, pe_dim=navi_rpe.out_dim,
                      out=prep["navi_pe"]) if pe_rides else None
            if fused_front:
                l0 = self.tf_ag2agmptl.layers[0]
                fp = front_proj_buffers(n * A, dev)
                (i_am, m_am, r_am, _), (i_aa, m_aa, r_aa, _), (i_at, m_at, r_at, _) = hip.front(
                    window=dict(attr=prep["attr"], pe=prep["pe"], row_invalid=prep["row_invalid"], in_images=wt_[0], pn_images=wt_[1], window=W, out=x),
                    proj=tile_proj_part(l0.norm_src, l0.attn_src, fp["qkv"], True, fp["kv16"]), rider=rider if use_rider else None, jobs=jobs,
                    pose_embed_job=None if use_rider else pj)  # (no rider: prep["navi_pe"] is read by the heads chain - it must be filled here)
                rider = None  # (ran in that launch)
            elif n * A < 4096:  # one launch for the three searches (the 4-waves-per-row form of the kernel)
                (i_am, m_am, r_am, _), (i_aa, m_aa, r_aa, _), (i_at, m_at, r_at, _) = hip.knn_embed_multi(jobs, pose_embed_job=pj)
            else:
                (i_am, m_am, r_am, _), (i_aa, m_aa, r_aa, _), (i_at, m_at, r_at, _) = (hip.knn_embed(**q) for q in jobs)
        knn_done = main.record_event() if knn_main else None

        def side_work():
            if knn_main:
                aux_stream.wait_event(knn_done)  # fork behind the searches' launch (the destination's pose embedding rides in it)
            with torch.cuda.stream(aux_stream if aux_stream is not None else main):
                if want_pe and not use_rider:
                    if not pe_rides:
                        prep["navi_pe"] = hip.pose_embed(prep["navi_pose3"], navi_rpe.pe_xy.freqs, navi_rpe.pe_yaw.freqs, navi_rpe.out_dim,
                                                         out=prep.get("navi_pe"))
                    if aux_tail is not None and not use_rider:
                        aux_tail(prep)

        if not knn_main:
            side_work()
        prep.update(knn_idx_ag2ag=i_aa, knn_invalid_ag2ag=m_aa, knn_idx_ag2mp=i_am, knn_invalid_ag2mp=m_am,
                    knn_idx_ag2tl=i_at, knn_invalid_ag2tl=m_at, _knn_aa=(i_aa, m_aa, r_aa), _knn_am=(i_am, m_am, r_am),
                    _knn_at=(i_at, m_at, r_at))
        kv_mp = self.kv_mp(mp)
        # (the searches are joined inside run_block, right before the first attention call: the first projection chain needs x only)
        # heads_tail(prep) -> the tbx_heads_tail_t fields (or None): the caller's heads in the last layer's launch; prep["_heads_done"] tells
        prep["_heads_done"] = run_block(self.tf_ag2agmptl, x, tok_inv, n, A, SelfKnn(i_aa, m_aa, rel=r_aa), heads_tail=None if heads_tail is None else (lambda **kw: heads_tail(prep, **kw)),
                  cross=lambda l: [Seg(kv_mp, l * 2 * D, l * 2 * D + D, M, i_am, m_am, None, mp_batch_div, rel=r_am),
                                   Seg(tl_kv, l * 2 * D, l * 2 * D + D, L, i_at, m_at, None, tl_batch_div, rel=r_at)], tail=tail, pose_rpe=rp,
                  join_stream=aux_stream, first_proj=fp, join_late=knn_main, after_first_proj=side_work if knn_main else None,
                  proj_rider=rider)
        return x, prep

    def _window_tile_images(self, attr_cols: int):
        """(input MLP images, PointNet images) for tbx_window_tile, or None where the module is not of the shape that kernel is
        built for (the default: "cat" input encoder 20 -> 64 -> 64 -> 64 + 64-d pose embedding, three 128 -> 64 PointNet layers)."""
        ie, te = self.input_encoder, self.temp_encoder
        lins = ie.mlp.linear_layers()
        if not (ie.mode == "cat" and len(lins) == 3 and ie.mlp.output_dim == 64 and ie.pe_dim == 64 and ie.mlp.input_dim <= 32
                and attr_cols >= 32 and all(ln is None for _, ln, _ in lins) and [a for _, _, a in lins] == [True, True, False]
                and len(te.mlp_layers) == 3):
            return None
        pn = []
        for mlp in te.mlp_layers:
            ll = mlp.linear_layers()
            if len(ll) != 1 or ll[0][1] is not None or tuple(ll[0][0].weight.shape) != (64, 128):
                return None
            pn.append(hip.packed_weight(ll[0][0].weight, ll[0][0].bias, mfma32=True))
        ins = [hip.packed_weight(hip.padded_weight(lins[0][0].weight, 32), lins[0][0].bias, mfma32=True)]
        ins += [hip.packed_weight(l.weight, l.bias, mfma32=True) for l, _, _ in lins[1:]]
        return ins, pn

    @staticmethod
    def pad_hist(ag_valid: Tensor, ag_pose: Tensor, ag_motion: Tensor, window: int):
        """Left-pad a [n,A,n_step,..] history (n_step <= window) with invalid steps: same tokens, fixed shape."""
        n, A, n_step = ag_valid.shape
        assert n_step <= window
        p = window - n_step
        hv = torch.cat([torch.zeros(n, A, p, dtype=torch.uint8, device=ag_valid.device), ag_valid.to(torch.uint8)], 2)
        hp = torch.cat([torch.zeros(n, A, p, 3, dtype=torch.float32, device=ag_valid.device), ag_pose.float()], 2)
        hm = torch.cat([torch.zeros(n, A, p, 3, dtype=torch.float32, device=ag_valid.device), ag_motion.float()], 2)
        return hv.contiguous(), hp.contiguous(), hm.contiguous()

    def forward(self, called_by_latent_encoder=False, **kw) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference signature (`_forward_hptr` kwargs): ag_valid [n,A,n_step], ag_attr [n,A,6], ag_motion, ag_pose,
        mp_token_*, tl_token_invalid / tl_token_feature / tl_token_pose."""
        n, A, _ = kw["ag_valid"].shape
        hv, hp, hm = self.pad_hist(kw["ag_valid"], kw["ag_pose"], kw["ag_motion"], self.temp_window_size)
        mp = {k: v for k, v in kw.items() if k.startswith("mp_token") or k.startswith("_kv")}
        tl_feat = kw["tl_token_feature"].reshape(-1, self.hidden_dim).contiguous().float()
        tl_kv = kv_tables(tl_feat, self.tl_kv_layers())
        x, _ = self.encode(hv, hp, hm, kw["ag_attr"].float().contiguous(), mp, kw["tl_token_invalid"].to(torch.uint8).contiguous(),
                           kw["tl_token_pose"].float().contiguous(), tl_kv)
        return x.view(n, A, self.hidden_dim), None
