"""ctypes binding of libtbx_hip.so (include/tbx_hip.h) + the `Chain` builder for tbx_rowchain programs.

The HIP library IS the product path: there is no CPU or PyTorch fallback. `load()` raises if the shared object is
missing, and every wrapper raises on a non-zero tbx return code or on tensors that are not on a HIP device.
"""
import ctypes as C
import os
import re
from pathlib import Path
from typing import List, Optional, Sequence

import torch

from . import abi, hip_base
from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import HEADER_PATH, LIB_PATH, load  # noqa: F401
from .hip_base import *  # noqa: F401,F403  (Seg, stream_ptr, packed_weight / stacked_linear / padded_weight)
from .hip_base import _check, _cptr, _drop_args, _ptr  # noqa: F401
from .hip_chain import *  # noqa: F401,F403  (Chain, group_tile_rows)
from .hip_rules import *  # noqa: F401,F403  (rule_tables, rule_check, rule_accumulate, filter_futures)
from .hip_train import *  # noqa: F401,F403  (the training entry points)
from .hip_train import _drop6  # noqa: F401


# ------------------------------------------------------------------------------------------------ wrappers
def knn_embed(src_pose, src_invalid, tgt_pose, tgt_invalid, k: int, dist_limit: float, freqs_xy=None, freqs_yaw=None,
              pe_dim: int = 128, tgt_batch_div: int = 1, want_rel_pose: bool = False, want_emb: bool = True, out=None):
    """-> idx i32 [n,S,k], invalid u8 [n,S,k], rel_pose f32 [n,S,k,3] | None, emb f32 [n,S,k,pe_dim] | None.
    out = (idx, invalid, rel_pose) of a previous call: written in place (buffers shared across streams / steps)."""
    n, S, _ = src_pose.shape
    T = tgt_pose.shape[1]
    dev = src_pose.device
    if out is not None:
        idx, inv, rel = out
        assert idx.shape == (n, S, k) and inv.shape == (n, S, k) and (rel is not None) == want_rel_pose
    else:
        idx = torch.empty(n, S, k, dtype=torch.int32, device=dev)
        inv = torch.empty(n, S, k, dtype=torch.uint8, device=dev)
        rel = torch.empty(n, S, k, 3, dtype=torch.float32, device=dev) if want_rel_pose else None
    emb = torch.empty(n, S, k, pe_dim, dtype=torch.float32, device=dev) if want_emb else None
    rc = load().tbx_knn_embed(_cptr(src_pose, torch.float32), _cptr(src_invalid, torch.uint8), _cptr(tgt_pose, torch.float32),
                              _cptr(tgt_invalid, torch.uint8), n, S, T, tgt_batch_div, k, float(dist_limit), _ptr(idx),
                              _ptr(inv), _ptr(rel), _ptr(emb), _cptr(freqs_xy), _cptr(freqs_yaw), pe_dim, stream_ptr())
    _check(rc, "tbx_knn_embed")
    return idx, inv, rel, emb


def knn_embed_multi(jobs, freqs_xy=None, freqs_yaw=None, pe_dim: int = 128, pose_embed_job=None):
    """Several `knn_embed` searches in one launch. jobs: dicts with knn_embed's arguments (src_pose, src_invalid, tgt_pose,
    tgt_invalid, k, dist_limit, tgt_batch_div, want_rel_pose, want_emb, out) -> list of (idx, invalid, rel_pose, emb).
    pose_embed_job = dict(pose3, freqs_xy, freqs_yaw, pe_dim, out[, col_off]): a `pose_embed` in the same launch (tbx_knn_embed_multi_pe)."""
    outs, cj = _knn_jobs(jobs, pe_dim)
    if pose_embed_job is not None:
        pj = _pose_job(pose_embed_job)
        _check(load().tbx_knn_embed_multi_pe(cj, len(jobs), _cptr(freqs_xy), _cptr(freqs_yaw), pe_dim, C.byref(pj), stream_ptr()),
               "tbx_knn_embed_multi_pe")
        return outs
    _check(load().tbx_knn_embed_multi(cj, len(jobs), _cptr(freqs_xy), _cptr(freqs_yaw), pe_dim, stream_ptr()), "tbx_knn_embed_multi")
    return outs


def _pose_job(q) -> "PoseEmbedJob":
    out = q["out"]
    return PoseEmbedJob(_cptr(q["pose3"], torch.float32), _cptr(q["freqs_xy"]), _cptr(q["freqs_yaw"]), _ptr(out, torch.float32),
                        q["pose3"].numel() // 3, int(q["pe_dim"]), out.stride(0), int(q.get("col_off", 0)), 0)


def _knn_jobs(jobs, pe_dim: int = 128):
    """-> ([(idx, invalid, rel_pose, emb)], tbx_knn_job_t array) for knn_embed_multi's job dicts."""
    outs, cj = [], (KnnJob * len(jobs))()
    for j, q in enumerate(jobs):
        n, S, _ = q["src_pose"].shape
        T, k, dev = q["tgt_pose"].shape[1], q["k"], q["src_pose"].device
        want_rel, want_emb = q.get("want_rel_pose", False), q.get("want_emb", True)
        if q.get("out") is not None:
            idx, inv, rel = q["out"]
            assert idx.shape == (n, S, k) and inv.shape == (n, S, k) and (rel is not None) == want_rel
        else:
            idx = torch.empty(n, S, k, dtype=torch.int32, device=dev)
            inv = torch.empty(n, S, k, dtype=torch.uint8, device=dev)
            rel = torch.empty(n, S, k, 3, dtype=torch.float32, device=dev) if want_rel else None
        emb = torch.empty(n, S, k, pe_dim, dtype=torch.float32, device=dev) if want_emb else None
        cj[j] = KnnJob(_cptr(q["src_pose"], torch.float32), _cptr(q["src_invalid"], torch.uint8), _cptr(q["tgt_pose"], torch.float32),
                       _cptr(q["tgt_invalid"], torch.uint8), _ptr(idx), _ptr(inv), _ptr(rel), _ptr(emb), n, S, T,
                       q.get("tgt_batch_div", 1), k, float(q["dist_limit"]))
        outs.append((idx, inv, rel, emb))
    return outs, cj


def rel_pose_dense(src_pose, src_invalid, tgt_pose, tgt_invalid, tgt_batch_div: int = 1, want_rel_pose: bool = True, want_dist: bool = True):
    """utils/rpe.py:8-58 as dense tensors -> rel_pose f32 [n,S,T,3] | None, rel_dist f32 [n,S,T] | None (+inf on invalid pairs)."""
    n, S, _ = src_pose.shape
    T = tgt_pose.shape[1]
    dev = src_pose.device
    rel = torch.empty(n, S, T, 3, dtype=torch.float32, device=dev) if want_rel_pose else None
    dist = torch.empty(n, S, T, dtype=torch.float32, device=dev) if want_dist else None
    _check(load().tbx_rel_pose_dense(_cptr(src_pose, torch.float32), _cptr(src_invalid, torch.uint8), _cptr(tgt_pose, torch.float32),
                                     _cptr(tgt_invalid, torch.uint8), n, S, T, tgt_batch_div, _ptr(rel), _ptr(dist), stream_ptr()),
           "tbx_rel_pose_dense")
    return rel, dist


def diffbar_reward(pred_valid, pred_pose, pred_motion, gt_valid, gt_pose, gt_motion, w_pos: float, w_rot: float, w_spd: float):
    """utils/rewards.py:35-85 for one step -> (out4 f32 [..., 4] = pos, rot, spd, sum; valid u8 [...]). gt_valid None: no imitation terms."""
    n = pred_valid.numel()
    out4 = torch.empty(*pred_valid.shape, 4, dtype=torch.float32, device=pred_pose.device)
    ov = torch.empty(pred_valid.shape, dtype=torch.uint8, device=pred_pose.device)
    has_gt = gt_valid is not None
    _check(load().tbx_diffbar_reward(_cptr(pred_valid, torch.uint8), _cptr(pred_pose, torch.float32), _cptr(pred_motion, torch.float32),
                                     _cptr(gt_valid, torch.uint8) if has_gt else None, _cptr(gt_pose, torch.float32) if has_gt else None,
                                     _cptr(gt_motion, torch.float32) if has_gt else None, n, float(w_pos), float(w_rot), float(w_spd),
                                     _ptr(out4), _ptr(ov), stream_ptr()), "tbx_diffbar_reward")
    return out4, ov


def pose_embed(pose3, freqs_xy, freqs_yaw, pe_dim: int, out=None, col_off: int = 0):
    n = pose3.numel() // 3
    if out is None:
        out = torch.empty(n, pe_dim, dtype=torch.float32, device=pose3.device)
    rc = load().tbx_pose_embed(_cptr(pose3, torch.float32), n, _cptr(freqs_xy), _cptr(freqs_yaw), pe_dim, _ptr(out),
                               out.stride(0), col_off, stream_ptr())
    _check(rc, "tbx_pose_embed")
    return out


def knarpe_attn(qbuf, q_off: int, qt_off: int, rpe_k_bias, n_batch: int, n_src: int, segs: Sequence[Seg], out, row_no_valid,
                freqs_xy=None, freqs_yaw=None, drop=None, fold=None):
    """drop = None, or (p, seed int64[1] device tensor, call id[, time_batch, time0]): attention-probability dropout
    (training; the last two for time-batched calls, include/tbx_hip.h).
    fold = the tbx_pack_weight_gemv image of linear_rpe's value half: tbx_knarpe_attn_fwd_folded, `out` is then [rows, >= 128]."""
    arr = (AttnSeg * len(segs))(*[s.c() for s in segs])
    if fold is not None:
        assert drop is None
        rc = load().tbx_knarpe_attn_fwd_folded(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, _ptr(rpe_k_bias, torch.float32),
                                               n_batch, n_src, arr, len(segs), _ptr(out, torch.float32), out.stride(0),
                                               _ptr(row_no_valid, torch.uint8), _cptr(freqs_xy), _cptr(freqs_yaw),
                                               _ptr(fold, torch.float32), stream_ptr())
        _check(rc, "tbx_knarpe_attn_fwd_folded")
        return
    p, seed, call, tb, t0 = _drop_args(drop)
    rc = load().tbx_knarpe_attn_fwd_dropout_tb(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, _ptr(rpe_k_bias, torch.float32),
                                            n_batch, n_src, arr, len(segs), _ptr(out, torch.float32), out.stride(0),
                                            _ptr(row_no_valid, torch.uint8), _cptr(freqs_xy), _cptr(freqs_yaw), float(p),
                                            _ptr(seed, torch.int64), int(call), tb, t0, stream_ptr())
    _check(rc, "tbx_knarpe_attn_fwd")


def knarpe_attn_mfma(qbuf, q_off: int, qt_off: int, n_batch: int, n_src: int, segs: Sequence[Seg], out, row_no_valid, freqs_xy, freqs_yaw,
                     drop=None):
    """tbx_knarpe_attn_fwd_mfma: the wave-per-row forward on the bf16 matrix cores (bf16 operands, fp32 accumulation / softmax).
    Same `out` [rows, >= 640] / row_no_valid as knarpe_attn; segments in the relative-pose form. drop: as knarpe_attn's (training:
    tbx_knarpe_attn_fwd_mfma_dropout_tb, the same mask as the VALU kernels draw for that key)."""
    arr = (AttnSeg * len(segs))(*[s.c() for s in segs])
    if drop is not None:
        p, seed, call, tb, t0 = _drop_args(drop)
        rc = load().tbx_knarpe_attn_fwd_mfma_dropout_tb(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, n_batch, n_src, arr, len(segs),
                                                        _ptr(out, torch.float32), out.stride(0), _ptr(row_no_valid, torch.uint8), _cptr(freqs_xy),
                                                        _cptr(freqs_yaw), float(p), _ptr(seed, torch.int64), int(call), tb, t0, stream_ptr())
        _check(rc, "tbx_knarpe_attn_fwd_mfma_dropout_tb")
        return
    rc = load().tbx_knarpe_attn_fwd_mfma(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, n_batch, n_src, arr, len(segs),
                                         _ptr(out, torch.float32), out.stride(0), _ptr(row_no_valid, torch.uint8), _cptr(freqs_xy),
                                         _cptr(freqs_yaw), stream_ptr())
    _check(rc, "tbx_knarpe_attn_fwd_mfma")


# ---- deferred launches (round 6: RolloutEngine's one-queue step). While a list is installed, tbx_front / tbx_knarpe_dec_layer calls
# are not launched but appended to it as (entry point name, descriptor, objects the descriptor points into); the engine then pairs the
# lights' list with the agents' and issues tbx_front_pair / tbx_knarpe_dec_layer_pair. Anything else launched meanwhile is an error of the
# schedule (the pairing assumes the two lists ARE the step), so the other wrappers refuse to run while a list is installed.
class defer:
    """with hip.defer() as calls: tbx_front / tbx_knarpe_dec_layer wrappers append (entry point, argument struct, keep-alive) to `calls`
    instead of launching (hip_base.DEFERRED); any other launch inside raises (hip_base.stream_ptr). The rollout engine's one-queue
    step collects the lights' and the agents' halves this way and launches them pairwise (launch_front_pair, launch_dec_layer_pair)."""

    def __enter__(self):
        assert hip_base.DEFERRED is None, "deferred launch lists do not nest"
        self.calls = []
        hip_base.DEFERRED = self.calls
        return self.calls

    def __exit__(self, *exc):
        hip_base.DEFERRED = None


def launch_front_pair(agents, lights) -> None:
    """agents / lights: ("tbx_front", Front, keep) entries of two deferred lists."""
    assert agents[0] == lights[0] == "tbx_front"
    _check(load().tbx_front_pair(C.byref(agents[1]), C.byref(lights[1]), stream_ptr()), "tbx_front_pair")


def launch_dec_layer_pair(a, b) -> None:
    assert a[0] == b[0] == "tbx_knarpe_dec_layer"
    _check(load().tbx_knarpe_dec_layer_pair(C.byref(a[1]), C.byref(b[1]), stream_ptr()), "tbx_knarpe_dec_layer_pair")


def launch_deferred(call) -> None:
    """One deferred entry as its own launch (the unpaired remainder of a list)."""
    _check(getattr(load(), call[0])(C.byref(call[1]), stream_ptr()), call[0])


def knarpe_dec_mid(qkv, q_off: int, qt_off: int, x, self_seg: Seg, cross_segs: Sequence[Seg], bias_self, bias_cross, ln, n_batch: int,
                   n_src: int, fold_self, out_proj, q_img, qfold, fold_cross, out2, flag2, freqs_xy=None, freqs_yaw=None, tail=None):
    """tbx_knarpe_dec_mid: folded self attention -> out-proj into x -> LN -> q -> W_k^T q -> folded cross attention, one launch.
    ln = (weight, bias, eps); the five images are tbx_pack_weight_gemv images (include/tbx_hip.h).
    tail = dict(out_proj2, linear1, linear2 (images), norm2 (w, b, eps), src_invalid, and for all but the last layer next_in_proj,
    next_qfold (images), next_norm (w, b, eps), qkv_out): tbx_knarpe_dec_layer - the rest of the layer in the same launch
    (out2 / flag2 may then be None)."""
    t = DecLayer() if tail is not None else None
    a = t.mid if t is not None else DecMid()
    a.qkv, a.x = _ptr(qkv, torch.float32), _cptr(x, torch.float32)
    a.self_seg = self_seg.c()
    cs = [sg.c() for sg in cross_segs]
    a.cross_seg[0], a.cross_seg[1] = cs[0], cs[-1]
    a.rpe_k_bias_self, a.rpe_k_bias_cross = _ptr(bias_self, torch.float32), _ptr(bias_cross, torch.float32)
    a.freqs_xy, a.freqs_yaw = _cptr(freqs_xy), _cptr(freqs_yaw)
    a.fold_self_image, a.out_proj_image, a.q_image = _ptr(fold_self), _ptr(out_proj), _ptr(q_img)
    a.qfold_image, a.fold_cross_image = _ptr(qfold), _ptr(fold_cross)
    a.ln_weight, a.ln_bias, a.ln_eps = _ptr(ln[0], torch.float32), _ptr(ln[1], torch.float32), float(ln[2])
    a.out2, a.flag2 = _ptr(out2, torch.float32), _ptr(flag2, torch.uint8)
    a.ld_qkv, a.q_off, a.qt_off, a.ld_out2 = qkv.stride(0), q_off, qt_off, (out2.stride(0) if out2 is not None else 0)
    a.n_cross, a.n_batch, a.n_src = len(cross_segs), n_batch, n_src
    if t is None:
        assert hip_base.DEFERRED is None, "tbx_knarpe_dec_mid inside a deferred step"
        _check(load().tbx_knarpe_dec_mid(C.byref(a), stream_ptr()), "tbx_knarpe_dec_mid")
        return
    t.out_proj2_image, t.linear1_image, t.linear2_image = _ptr(tail["out_proj2"]), _ptr(tail["linear1"]), _ptr(tail["linear2"])
    n2 = tail["norm2"]
    t.norm2_weight, t.norm2_bias, t.norm2_eps = _ptr(n2[0], torch.float32), _ptr(n2[1], torch.float32), float(n2[2])
    t.src_invalid = _cptr(tail["src_invalid"], torch.uint8)
    t.tail_mfma32 = int(tail.get("mfma32") or 0)  # (1: three bf16 products per fp32 product; 2: one - Schedule.linear_bf16)
    qo = tail.get("qkv_out")
    if qo is not None:
        n3 = tail["next_norm"]
        t.next_in_proj_image, t.next_qfold_image = _ptr(tail["next_in_proj"]), _ptr(tail["next_qfold"])
        t.next_norm_weight, t.next_norm_bias, t.next_norm_eps = _ptr(n3[0], torch.float32), _ptr(n3[1], torch.float32), float(n3[2])
        t.qkv_out, t.ld_qkv_out = _ptr(qo, torch.float32), qo.stride(0)
        if tail.get("kv16_out") is not None:
            assert tail["kv16_out"].shape[1] == 256 and tail["kv16_out"].is_contiguous()
            t.kv16_out = _ptr(tail["kv16_out"], torch.bfloat16)
    hd, keep = tail.get("heads"), None
    if hd is not None:  # dict(images = 9 gemv images, navi_emb, latent_emb, navi_valid, latent_invalid, type_mask [3, rows] u8, action_out [rows, 2])
        keep = HeadsTail()
        for i, im in enumerate(hd["images"]):
            keep.images[i] = _ptr(im, torch.float32)
        keep.navi_emb, keep.latent_emb = _cptr(hd["navi_emb"], torch.float32), _cptr(hd["latent_emb"], torch.float32)
        keep.navi_valid, keep.latent_invalid = _cptr(hd["navi_valid"], torch.uint8), _cptr(hd["latent_invalid"], torch.uint8)
        keep.type_mask, keep.action_out = _cptr(hd["type_mask"], torch.uint8), _cptr(hd["action_out"], torch.float32)
        keep.mask_stride = hd["type_mask"].shape[1]
        if hd.get("sim_state") is not None:  # the fused step tail: (SimState, parts) + the next step's AgentPrepArgs
            keep.sim_state, keep.sim_parts = C.addressof(hd["sim_state"]), int(hd["sim_parts"])
            keep.next_prep = C.addressof(hd["next_prep"])
        t.heads = C.addressof(keep)
    lt_, keep_l = tail.get("lights"), None
    if lt_ is not None:
        keep_l = tl_tail_struct(lt_, x.shape[0])
        t.lights = C.addressof(keep_l)
    if hip_base.DEFERRED is not None:
        hip_base.DEFERRED.append(("tbx_knarpe_dec_layer", t, (keep, keep_l, tail, cs)))
        return
    _check(load().tbx_knarpe_dec_layer(C.byref(t), stream_ptr()), "tbx_knarpe_dec_layer")


def tl_tail_struct(lt_: dict, rows: int) -> "TlTail":
    """tbx_tl_tail_t from dict(kv_images[4], norms [(w, b, eps)] * 4, kv_out, mlp_images[3], tl_invalid, logits_out, clamp)."""
    keep_l = TlTail()
    for i in range(4):
        keep_l.kv_images[i] = _ptr(lt_["kv_images"][i], torch.float32)
        w_, b_, e_ = lt_["norms"][i]
        keep_l.norm_weight[i], keep_l.norm_bias[i], keep_l.norm_eps[i] = _ptr(w_, torch.float32), _ptr(b_, torch.float32), float(e_)
    for i in range(3):
        keep_l.mlp_images[i] = _ptr(lt_["mlp_images"][i], torch.float32)
    kvo = lt_["kv_out"]
    assert kvo.dim() == 2 and kvo.stride(1) == 1 and kvo.shape[0] == rows
    keep_l.kv_out, keep_l.ld_kv, keep_l.kv_bf16 = kvo.data_ptr(), kvo.stride(0), int(kvo.dtype == torch.bfloat16)
    assert kvo.dtype in (torch.bfloat16, torch.float32)
    lo = lt_["logits_out"]
    assert lo.is_contiguous() and lo.shape[0] == rows
    keep_l.tl_invalid, keep_l.logits_out, keep_l.n_state = _cptr(lt_["tl_invalid"], torch.uint8), _ptr(lo, torch.float32), lo.shape[1]
    keep_l.clamp_lo, keep_l.clamp_hi = (float(v) for v in lt_["clamp"])
    sim = lt_.get("sim")  # dict(state = SimState of the lights, parts, attr, row_invalid): the light's own step behind its logits
    if sim is not None:
        keep_l.sim_state, keep_l.sim_parts = C.addressof(sim["state"]), int(sim["parts"])
        keep_l.prep_attr, keep_l.prep_ld_attr = _ptr(sim["attr"], torch.float32), sim["attr"].stride(0)
        keep_l.prep_row_invalid = _ptr(sim["row_invalid"], torch.uint8)
    return keep_l


def tl_tail_tile(x: torch.Tensor, lights: dict) -> None:
    """tbx_tl_tail_tile (tbx_tl_tail_tile_bf16 under Schedule.linear_bf16) on the finished light tokens x [rows, 128]."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == 128
    st = tl_tail_struct(lights, x.shape[0])
    fn = "tbx_tl_tail_tile_bf16" if tile_products() == 1 else "tbx_tl_tail_tile"
    _check(getattr(load(), fn)(_ptr(x, torch.float32), x.shape[0], C.byref(st), stream_ptr()), fn)


def _layer_tile_args(x, attn=None, ffn=None, proj=None, store_x: bool = True, drop=None, rider=None) -> "LayerTile":
    """tbx_layer_tile on the token rows x [rows, 128] (in place). Each part is None or a dict:
    attn = dict(out [rows, >= 640], row_no_valid u8 [rows], fold, out_proj (mfma32 images));
    ffn = dict(norm2 (w, b, eps), linear1, linear2 (images), src_invalid u8 [rows] | None);
    proj = dict(norm (w, b, eps), image, qfold (images), n = 128 | 384, out [rows, >= 640 | 896], kv16 = bf16 [rows, 256] | None);
    drop = the keyed dropouts of training's stepping pass (see below);
    rider = dict(inp [r, 128] | pose3 [r, 3] + freqs, add, out [r, 128], valid u8 [r], images = 4 mfma32 images): tbx_layer_tile_t's rider_* (a
    first-projection launch only)."""
    a = LayerTile()
    a.x, a.n_rows, a.store_x = _cptr(x, torch.float32), x.shape[0], int(store_x)
    assert x.dim() == 2 and x.shape[1] == 128
    if attn is not None:
        o = attn["out"]
        assert o.dim() == 2 and o.stride(1) == 1 and o.shape[0] == x.shape[0]
        a.attn_out, a.ld_attn, a.row_no_valid = _ptr(o, torch.float32), o.stride(0), _cptr(attn["row_no_valid"], torch.uint8)
        a.fold_image, a.out_proj_image = _ptr(attn["fold"], torch.float32), _ptr(attn["out_proj"], torch.float32)
    if ffn is not None:
        w, b, eps = ffn["norm2"]
        a.norm2_weight, a.norm2_bias, a.norm2_eps = _cptr(w, torch.float32), _cptr(b, torch.float32), float(eps)
        a.linear1_image, a.linear2_image = _ptr(ffn["linear1"], torch.float32), _ptr(ffn["linear2"], torch.float32)
        a.src_invalid = _cptr(ffn.get("src_invalid"), torch.uint8)
    if proj is not None:
        w, b, eps = proj["norm"]
        a.proj_norm_weight, a.proj_norm_bias, a.proj_norm_eps = _cptr(w, torch.float32), _cptr(b, torch.float32), float(eps)
        a.proj_image, a.qfold_image, a.proj_n = _ptr(proj["image"], torch.float32), _ptr(proj["qfold"], torch.float32), int(proj["n"])
        o = proj["out"]
        assert o.dim() == 2 and o.stride(1) == 1 and o.shape[0] == x.shape[0]
        a.proj_out, a.ld_proj = _ptr(o, torch.float32), o.stride(0)
        kv16 = proj.get("kv16")
        if kv16 is not None:
            assert kv16.shape == (x.shape[0], 256) and kv16.is_contiguous()
            a.kv16_out = _ptr(kv16, torch.bfloat16)
    if drop is not None:  # dict(p, seed int64[1] device tensor, step, sites = (attention residual, FFN hidden, FFN output) | None each)
        th = drop["p"] * 4294967296.0
        a.drop_thresh = 1 if 0 < th < 1 else int(th)
        a.drop_scale = 1.0 / (1.0 - drop["p"])
        a.drop_seed, a.drop_step = _ptr(drop["seed"], torch.int64), int(drop["step"])
        for i, st in enumerate(drop["sites"]):
            a.drop_site[i] = -1 if st is None else int(st)
    if rider is not None:
        r = rider["out"].shape[0]
        for k in ("add", "out") + (("inp",) if rider.get("pose3") is None else ()):
            assert rider[k].shape == (r, 128) and rider[k].is_contiguous()
        a.rider_add, a.rider_out = _ptr(rider["add"], torch.float32), _ptr(rider["out"], torch.float32)
        if rider.get("pose3") is not None:  # stage 0 on the pose embedding of pose3 [r, 3], built in the kernel (freqs = (xy, yaw) tables)
            assert rider["pose3"].shape == (r, 3)
            a.rider_pose3 = _cptr(rider["pose3"], torch.float32)
            a.rider_freqs_xy, a.rider_freqs_yaw = _cptr(rider["freqs"][0], torch.float32), _cptr(rider["freqs"][1], torch.float32)
        else:
            a.rider_in = _ptr(rider["inp"], torch.float32)
        assert rider["valid"].numel() == r and len(rider["images"]) == 4
        a.rider_valid, a.rider_rows = _cptr(rider["valid"], torch.uint8), r
        for i, im in enumerate(rider["images"]):
            a.rider_images[i] = _ptr(im, torch.float32)
    return a


# products per LINEAR of the tile kernels: 3 (fp32-class) or 1 (the *_bf16 entry points); engine.py points this at the scoped
# Schedule (linear_bf16) - this module knows no schedules
tile_products = lambda: 3


def layer_tile(x, attn=None, ffn=None, proj=None, store_x: bool = True, drop=None, rider=None):
    """tbx_layer_tile (arguments: _layer_tile_args)."""
    a = _layer_tile_args(x, attn, ffn, proj, store_x, drop, rider)
    fn = "tbx_layer_tile_bf16" if (tile_products() == 1 and drop is None) else "tbx_layer_tile"
    _check(getattr(load(), fn)(C.byref(a), stream_ptr()), fn)


def heads_tile(x, hd: dict):
    """tbx_heads_tile: hd = dict(images = 9 mfma32 images, navi_emb, latent_emb [rows, 128], navi_valid, latent_invalid u8 [rows],
    type_mask u8 [3, rows], action_out [rows, 2])."""
    a = HeadsTile()
    a.x, a.n_rows = _cptr(x, torch.float32), x.shape[0]
    raw = hd.get("raw")
    if raw is not None:  # dict(navi_pe, dest_feature [rows, 128], latent_z [rows, >= 16], images = 7 mfma32 images, drop = None | dict(p, seed, step, sites[12]))
        a.raw = 1
        a.navi_pe, a.dest_feature = _cptr(raw["navi_pe"], torch.float32), _cptr(raw["dest_feature"], torch.float32)
        z = raw["latent_z"]
        assert z.dim() == 2 and z.stride(1) == 1 and z.shape[1] >= 16 and z.shape[0] == x.shape[0]
        a.latent_z, a.ld_z = _ptr(z, torch.float32), z.stride(0)
        assert len(raw["images"]) == 7
        for i, im in enumerate(raw["images"]):
            a.raw_images[i] = _ptr(im, torch.float32)
        dr = raw.get("drop")
        if dr is not None:
            th = dr["p"] * 4294967296.0
            a.drop_thresh, a.drop_scale = (1 if 0 < th < 1 else int(th)), 1.0 / (1.0 - dr["p"])
            a.drop_seed, a.drop_step = _ptr(dr["seed"], torch.int64), int(dr["step"])
            for i, st in enumerate(dr["sites"]):
                a.drop_site[i] = -1 if st is None else int(st)
    else:
        a.navi_emb, a.latent_emb = _cptr(hd["navi_emb"], torch.float32), _cptr(hd["latent_emb"], torch.float32)
    a.navi_valid, a.latent_invalid = _cptr(hd["navi_valid"], torch.uint8), _cptr(hd["latent_invalid"], torch.uint8)
    a.type_mask, a.mask_stride, a.action_out = _cptr(hd["type_mask"], torch.uint8), hd["type_mask"].shape[1], _cptr(hd["action_out"], torch.float32)
    assert len(hd["images"]) == 9 and hd["action_out"].shape == (x.shape[0], 2)
    for i, im in enumerate(hd["images"]):
        a.images[i] = _ptr(im, torch.float32)
    fn = "tbx_heads_tile_bf16" if (tile_products() == 1 and a.drop_thresh == 0) else "tbx_heads_tile"
    _check(getattr(load(), fn)(C.byref(a), stream_ptr()), fn)


def _window_tile_args(attr, pe, row_invalid, in_images, pn_images, window: int, out, add_mode: bool = False, drop=None) -> "WindowTile":
    """tbx_window_tile. cat mode: attr [G * window, >= 4 cols], pe [G * window, 64]; add mode: pe = one feature row per window
    [G, 128], the input MLP is 128 wide. row_invalid u8 [G * window] -> out [G, 128]."""
    a = WindowTile()
    assert attr.dim() == 2 and attr.stride(1) == 1 and pe.is_contiguous() and out.shape[1] == 128 and out.is_contiguous()
    assert pe.shape == ((out.shape[0], 128) if add_mode else (attr.shape[0], 64))
    a.attr_cols, a.d_mlp, a.add_mode = min(32, attr.shape[1] // 4 * 4), (128 if add_mode else 64), int(add_mode)
    a.attr, a.ld_attr, a.pe, a.row_invalid = _ptr(attr, torch.float32), attr.stride(0), _ptr(pe, torch.float32), _cptr(row_invalid, torch.uint8)
    for i in range(3):
        a.in_images[i], a.pn_images[i] = _ptr(in_images[i], torch.float32), _ptr(pn_images[i], torch.float32)
    a.out, a.window, a.n_groups = _ptr(out, torch.float32), int(window), out.shape[0]
    assert attr.shape[0] == out.shape[0] * window
    if drop is not None:  # dict(p, seed, step, sites = the three PointNet layers' dropout site ids): training's stepping pass
        th = drop["p"] * 4294967296.0
        a.drop_thresh, a.drop_scale = (1 if 0 < th < 1 else int(th)), 1.0 / (1.0 - drop["p"])
        a.drop_seed, a.drop_step = _ptr(drop["seed"], torch.int64), int(drop["step"])
        for i, st in enumerate(drop["sites"]):
            a.drop_site[i] = int(st)
    return a


def window_tile(attr, pe, row_invalid, in_images, pn_images, window: int, out, add_mode: bool = False, drop=None):
    """tbx_window_tile (arguments: _window_tile_args)."""
    a = _window_tile_args(attr, pe, row_invalid, in_images, pn_images, window, out, add_mode, drop)
    fn = "tbx_window_tile_bf16" if (tile_products() == 1 and drop is None) else "tbx_window_tile"
    _check(getattr(load(), fn)(C.byref(a), stream_ptr()), fn)


def front(window: dict, proj: dict, rider=None, jobs=None, pose_embed_job=None):
    """tbx_front: the window PointNet (window = window_tile's arguments as a dict), the first projection of its pooled rows (proj =
    layer_tile's proj dict) + rider, and the K-nearest searches `jobs` (knn_embed_multi's job dicts, relative-pose form) + pose-
    embedding job in ONE launch. -> the searches' [(idx, invalid, rel_pose, None)]."""
    f = Front()
    f.win = _window_tile_args(**window)
    f.layer = _layer_tile_args(window["out"], proj=proj, store_x=False, rider=rider)
    outs, keep = [], []
    if jobs:
        assert all(not q.get("want_emb", True) for q in jobs)
        outs, cj = _knn_jobs(jobs)
        keep.append(cj)
        f.jobs, f.n_jobs, f.pe_dim = C.addressof(cj), len(jobs), 128
        if pose_embed_job is not None:
            pj = _pose_job(pose_embed_job)
            keep.append(pj)
            f.pe = C.addressof(pj)
    if hip_base.DEFERRED is not None:
        hip_base.DEFERRED.append(("tbx_front", f, (keep, window, proj, rider, jobs, outs)))
        return outs
    _check(load().tbx_front(C.byref(f), stream_ptr()), "tbx_front")
    return outs


def agent_prep_args(hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, freqs_xy, freqs_yaw, pe_dim, out, dest=None,
                    mp_tok_pose=None, n_mp=0, mp_batch_div=1) -> AgentPrepArgs:
    """agent_prep's arguments as tbx_agent_prep_args_t (the fused step tail of tbx_knarpe_dec_layer)."""
    n, A, W = hist_valid.shape
    a = AgentPrepArgs()
    a.hist_valid, a.hist_pose, a.hist_motion = _cptr(hist_valid, torch.uint8), _cptr(hist_pose, torch.float32), _cptr(hist_motion, torch.float32)
    a.ag_attr6, a.ag_type_idx = _cptr(ag_attr6, torch.float32), _cptr(ag_type_idx, torch.uint8)
    a.freqs_xy, a.freqs_yaw = _cptr(freqs_xy), _cptr(freqs_yaw)
    a.tok_pose, a.tok_invalid, a.attr, a.pe, a.row_invalid = (_ptr(out[k]) for k in ("tok_pose", "tok_invalid", "attr", "pe", "row_invalid"))
    a.type_mask, a.dest, a.mp_tok_pose = _ptr(out.get("type_mask")), _cptr(dest, torch.int64), _cptr(mp_tok_pose, torch.float32)
    a.navi_pose3, a.navi_row = _ptr(out.get("navi_pose3")), _ptr(out.get("navi_row"))
    a.n_tok, a.n_ag, a.window, a.pe_dim, a.n_mp, a.mp_batch_div = n * A, A, W, pe_dim, n_mp, mp_batch_div
    return a


def agent_prep(hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, freqs_xy, freqs_yaw, pe_dim, out, dest=None,
               mp_tok_pose=None, n_mp=0, mp_batch_div=1):
    n, A, W = hist_valid.shape
    rc = load().tbx_agent_prep(
        _cptr(hist_valid, torch.uint8), _cptr(hist_pose, torch.float32), _cptr(hist_motion, torch.float32),
        _cptr(ag_attr6, torch.float32), _cptr(ag_type_idx, torch.uint8), n, A, W, _cptr(freqs_xy), _cptr(freqs_yaw), pe_dim,
        _ptr(out["tok_pose"]), _ptr(out["tok_invalid"]), _ptr(out["attr"]), _ptr(out["pe"]), _ptr(out["row_invalid"]),
        _ptr(out.get("type_mask")), _cptr(dest, torch.int64), _cptr(mp_tok_pose, torch.float32), n_mp, mp_batch_div,
        _ptr(out.get("navi_pose3")), _ptr(out.get("navi_row")), stream_ptr())
    _check(rc, "tbx_agent_prep")


def tl_prep(hist_tl, tl_invalid, attr, row_invalid):
    n, L, W = hist_tl.shape
    rc = load().tbx_tl_prep(_cptr(hist_tl, torch.uint8), _cptr(tl_invalid, torch.uint8), n, L, W, attr.stride(0),
                            _ptr(attr, torch.float32), _ptr(row_invalid, torch.uint8), stream_ptr())
    _check(rc, "tbx_tl_prep")


def map_prep(mp_valid_u8, mp_type11, mp_pose, attr, pe, row_invalid, tok_pose, tok_invalid):
    n, M, N = mp_valid_u8.shape
    rc = load().tbx_map_prep(_cptr(mp_valid_u8, torch.uint8), _cptr(mp_type11, torch.float32), _cptr(mp_pose, torch.float32),
                             n, M, N, _ptr(attr), _ptr(pe), _ptr(row_invalid), _ptr(tok_pose), _ptr(tok_invalid), stream_ptr())
    _check(rc, "tbx_map_prep")


SIM_AGENTS, SIM_LIGHTS, SIM_ADVANCE, SIM_NO_DISABLE, SIM_NO_APPEND, SIM_APPEND = 1, 2, 4, 8, 16, 32


def sim_step(state: SimState, parts: int = SIM_AGENTS | SIM_LIGHTS | SIM_ADVANCE, tl_prep=None):
    """tl_prep = (tl_invalid u8 [n*L], attr f32 [n*L*W, ld], row_invalid u8 [n*L*W]): the lights' update also writes the tbx_tl_prep
    rows of their new windows (tbx_sim_step_tl_prep)."""
    assert hip_base.DEFERRED is None, "tbx_sim_step inside a deferred step (the one-queue step runs the lights' update in their last layer's tail)"
    if tl_prep is not None:
        inv, attr, row_inv = tl_prep
        _check(load().tbx_sim_step_tl_prep(C.byref(state), parts, _cptr(inv, torch.uint8), attr.stride(0), _ptr(attr, torch.float32),
                                           _ptr(row_inv, torch.uint8), stream_ptr()), "tbx_sim_step_tl_prep")
        return
    _check(load().tbx_sim_step_parts(C.byref(state), parts, stream_ptr()), "tbx_sim_step_parts")


