"""The C-ABI mirror: constants and ctypes structures of include/tbx_hip.h, and `load()` - dlopen of the in-tree libtbx_hip.so with
every declared symbol checked and every prototype set. Nothing here computes; the wrappers that launch kernels are in hip.py
(which re-exports this module's names). tests/test_abi_and_host.py compiles the header with gcc and compares every structure's
`sizeof` and every field's `offsetof` with the classes below."""
import ctypes as C
import os
import re
from pathlib import Path
from typing import List

_lib = None
_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "csrc" / "libtbx_hip.so"
HEADER_PATH = _PKG.parent / "include" / "tbx_hip.h"

# ---- constants mirrored from include/tbx_hip.h
OP_LOAD, OP_LINEAR, OP_LAYERNORM, OP_ADD, OP_COPY, OP_ROWMASK, OP_GROUPMAX, OP_POOLMAX, OP_STORE, OP_CLAMP, OP_DROPOUT = range(1, 12)
ACT_NONE, ACT_RELU = 0, 1
F_ACCUM, F_WT, F_ROW_DIV, F_ROW_MOD, F_ROW_IDX, F_ROW_BATCH_MOD, F_WPACK, F_MASK_INV = 1, 2, 4, 8, 16, 32, 64, 128
F_POOL_KEEP = 256
F_WSPLIT = 512
F_ROWSKIP = 1024
F_LOAD2 = 2048
F_WGEMV = 4096
F_OUT_BF16 = 8192
F_MASKED_SUM = 16384
F_ROWZERO = 32768
BUF0, BUF1, AUX, GLOBAL = 0, 1, 2, 3
MAX_STAGES, AUX_LD = 44, 260


class Stage(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("op", "src", "dst", "src_col", "dst_col", "k", "n", "act", "flags", "ld", "div",
                                         "reserved", "ld2", "pad")] + [("f0", C.c_float), ("f1", C.c_float), ("p0", C.c_void_p),
                                                                       ("p1", C.c_void_p), ("p2", C.c_void_p)]


class AttnSeg(C.Structure):
    _fields_ = [("kv", C.c_void_p), ("idx", C.c_void_p), ("invalid", C.c_void_p), ("emb", C.c_void_p), ("rel_pose", C.c_void_p)] + [
        (n, C.c_int32) for n in ("ld_kv", "k_off", "v_off", "n_tgt", "batch_div", "k", "kv_bf16")]


class KnnJob(C.Structure):
    """tbx_knn_job_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("src_pose", "src_invalid", "tgt_pose", "tgt_invalid", "idx", "invalid", "rel_pose", "emb")]
                + [(n, C.c_int32) for n in ("n_batch", "n_src", "n_tgt", "tgt_batch_div", "k")] + [("dist_limit", C.c_float)])


class PoseEmbedJob(C.Structure):
    """tbx_pose_embed_job_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("pose3", "freqs_xy", "freqs_yaw", "out")] + [("n", C.c_int64)]
                + [(n, C.c_int32) for n in ("pe_dim", "ld_out", "col_off", "reserved")])


class DecMid(C.Structure):
    """tbx_dec_mid_t (include/tbx_hip.h)."""
    _fields_ = ([("qkv", C.c_void_p), ("x", C.c_void_p), ("self_seg", AttnSeg), ("cross_seg", AttnSeg * 2)]
                + [(n, C.c_void_p) for n in ("rpe_k_bias_self", "rpe_k_bias_cross", "freqs_xy", "freqs_yaw", "fold_self_image",
                                             "out_proj_image", "q_image", "qfold_image", "fold_cross_image", "ln_weight", "ln_bias",
                                             "out2", "flag2")]
                + [("ln_eps", C.c_float)]
                + [(n, C.c_int32) for n in ("ld_qkv", "q_off", "qt_off", "ld_out2", "n_cross", "n_batch", "n_src")])


class HeadsTail(C.Structure):
    """tbx_heads_tail_t (include/tbx_hip.h)."""
    _fields_ = ([("images", C.c_void_p * 9)]
                + [(n, C.c_void_p) for n in ("navi_emb", "latent_emb", "navi_valid", "latent_invalid", "type_mask", "action_out")]
                + [("mask_stride", C.c_int32), ("sim_parts", C.c_int32), ("sim_state", C.c_void_p), ("next_prep", C.c_void_p)])


class AgentPrepArgs(C.Structure):
    """tbx_agent_prep_args_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("hist_valid", "hist_pose", "hist_motion", "ag_attr6", "ag_type_idx", "freqs_xy", "freqs_yaw", "tok_pose",
                                           "tok_invalid", "attr", "pe", "row_invalid", "type_mask", "dest", "mp_tok_pose", "navi_pose3", "navi_row")]
                + [(n, C.c_int32) for n in ("n_tok", "n_ag", "window", "pe_dim", "n_mp", "mp_batch_div")])


class DecLayer(C.Structure):
    """tbx_dec_layer_t (include/tbx_hip.h)."""
    _fields_ = ([("mid", DecMid)]
                + [(n, C.c_void_p) for n in ("out_proj2_image", "linear1_image", "linear2_image", "next_in_proj_image", "next_qfold_image",
                                             "norm2_weight", "norm2_bias", "next_norm_weight", "next_norm_bias", "src_invalid", "qkv_out", "kv16_out", "heads")]
                + [("norm2_eps", C.c_float), ("next_norm_eps", C.c_float), ("ld_qkv_out", C.c_int32), ("tail_mfma32", C.c_int32),
                   ("lights", C.c_void_p)])


class TlTail(C.Structure):
    """tbx_tl_tail_t (include/tbx_hip.h)."""
    _fields_ = [("kv_images", C.c_void_p * 4), ("norm_weight", C.c_void_p * 4), ("norm_bias", C.c_void_p * 4), ("norm_eps", C.c_float * 4),
                ("kv_out", C.c_void_p), ("mlp_images", C.c_void_p * 3), ("tl_invalid", C.c_void_p), ("logits_out", C.c_void_p),
                ("ld_kv", C.c_int32), ("kv_bf16", C.c_int32), ("n_state", C.c_int32), ("pad_", C.c_int32),
                ("clamp_lo", C.c_float), ("clamp_hi", C.c_float), ("sim_parts", C.c_int32), ("prep_ld_attr", C.c_int32),
                ("sim_state", C.c_void_p), ("prep_attr", C.c_void_p), ("prep_row_invalid", C.c_void_p)]


class PackJob(C.Structure):
    """tbx_pack_job_t (include/tbx_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32), ("ld", C.c_int32),
                ("groups", C.c_int32), ("wt", C.c_int32), ("pad_", C.c_int32)]


class LayerTile(C.Structure):
    """tbx_layer_tile_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("x", "attn_out", "row_no_valid", "fold_image", "out_proj_image", "norm2_weight", "norm2_bias",
                                           "linear1_image", "linear2_image", "src_invalid", "proj_norm_weight", "proj_norm_bias", "proj_image",
                                           "qfold_image", "proj_out", "kv16_out")]
                + [("norm2_eps", C.c_float), ("proj_norm_eps", C.c_float)]
                + [(n, C.c_int32) for n in ("ld_attn", "ld_proj", "proj_n", "store_x")] + [("n_rows", C.c_int64)]
                + [("drop_seed", C.c_void_p), ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float), ("drop_site", C.c_int32 * 3),
                   ("drop_step", C.c_int32)]
                + [("rider_in", C.c_void_p), ("rider_add", C.c_void_p), ("rider_pose3", C.c_void_p), ("rider_freqs_xy", C.c_void_p),
                   ("rider_freqs_yaw", C.c_void_p), ("rider_images", C.c_void_p * 4), ("rider_valid", C.c_void_p),
                   ("rider_out", C.c_void_p), ("rider_rows", C.c_int64)])


class HeadsTile(C.Structure):
    """tbx_heads_tile_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("x", "navi_emb", "latent_emb", "navi_valid", "latent_invalid", "type_mask")]
                + [("images", C.c_void_p * 9), ("action_out", C.c_void_p), ("mask_stride", C.c_int32), ("raw", C.c_int32), ("n_rows", C.c_int64),
                   ("navi_pe", C.c_void_p), ("dest_feature", C.c_void_p), ("latent_z", C.c_void_p), ("raw_images", C.c_void_p * 7),
                   ("drop_seed", C.c_void_p), ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float), ("drop_site", C.c_int32 * 12),
                   ("drop_step", C.c_int32), ("ld_z", C.c_int32)])


class WindowTile(C.Structure):
    """tbx_window_tile_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("attr", "pe", "row_invalid")] + [("in_images", C.c_void_p * 3), ("pn_images", C.c_void_p * 3),
                ("out", C.c_void_p), ("window", C.c_int32), ("ld_attr", C.c_int32), ("n_groups", C.c_int64)]
                + [(n, C.c_int32) for n in ("attr_cols", "d_mlp", "add_mode", "pad_")]
                + [("drop_seed", C.c_void_p), ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float), ("drop_site", C.c_int32 * 3),
                   ("drop_step", C.c_int32)])


class Front(C.Structure):
    """tbx_front_t (include/tbx_hip.h)."""
    _fields_ = [("win", WindowTile), ("layer", LayerTile), ("jobs", C.c_void_p), ("pe", C.c_void_p), ("freqs_xy", C.c_void_p),
                ("freqs_yaw", C.c_void_p), ("n_jobs", C.c_int32), ("pe_dim", C.c_int32)]


class SimState(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("n_batch", "n_ag", "n_tl", "window", "n_step_gt", "n_step_tl_gt", "n_step_out", "n_node")]
        + [(n, C.c_void_p) for n in (
            "step", "ag_valid", "ag_disabled", "ag_pose", "ag_motion", "navi_valid", "outside_map", "dest_reached",
            "tl_state", "hist_valid", "hist_pose", "hist_motion", "hist_tl", "ag_type_idx", "tf_mask", "gt_valid", "gt_pose",
            "gt_motion", "tl_gt", "boundary", "dest_pos", "dest_dir", "dest_invalid", "dest_kind", "dest_thresh",
            "action_mean", "tl_logits", "out_valid", "out_pose", "out_motion", "out_action", "out_tl_state",
            "out_outside_map", "out_dest_reached")]
        + [("max_acc", C.c_float * 3), ("max_yaw_rate", C.c_float * 3), ("dt", C.c_float)]
        + [(n, C.c_void_p) for n in ("out_reward", "out_reward_valid", "out_tf", "out_tl_nll")]
        + [(n, C.c_float) for n in ("w_pos", "w_rot", "w_spd")]
        + [(n, C.c_void_p) for n in ("player_valid", "player_action", "ov_valid", "ov_pose", "ov_motion", "ov_tl_valid",
                                     "ov_tl_state", "now_outside", "now_reached")]
    )


class TrainChainArgs(C.Structure):
    """tbx_train_chain_t (include/tbx_hip.h)."""
    _fields_ = ([(n, C.c_int32) for n in ("n_batch", "n_ag", "n_step", "n_step_gt", "n_node", "window")]
                + [(n, C.c_float) for n in ("dt", "w_pos", "w_rot", "w_spd")]
                + [(n, C.c_void_p) for n in (
                    "gt_valid", "gt_pose", "gt_motion", "tf_mask", "lim", "dest_pos", "dest_dir", "dest_invalid", "dest_thresh",
                    "dest_kind", "boundary", "valid", "disabled", "navi_valid", "outside", "reached", "pose", "motion",
                    "rec_valid", "rec_pose", "rec_motion", "rec_navi_valid", "pred_valid", "tf", "ov", "reward_valid",
                    "pred_pose", "pred_motion", "reward")])


class RuleCtx(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("n_batch", "n_ag", "n_tl", "map_batch_div", "cap")]
                + [(n, C.c_void_p) for n in ("seg", "n_seg", "lane", "n_lane", "ag_size", "ag_type_idx", "tl_valid", "tl_pose")]
                + [("collision_size_scale", C.c_float)]
                + [(n, C.c_void_p) for n in ("seg_start", "seg_grid", "lane_start", "lane_grid")])


RULE_COLLIDED, RULE_COLLIDED_WOSAC, RULE_RUN_ROAD_EDGE, RULE_RUN_RED_LIGHT, RULE_PASSIVE = 1, 2, 4, 8, 16

_lib = None


def declared_symbols() -> List[str]:
    """Entry points declared in include/tbx_hip.h (the C-ABI contract)."""
    txt = HEADER_PATH.read_text()
    return sorted(set(re.findall(r"^(?:int|int64_t|const char\*)\s+(tbx_\w+)\s*\(", txt, flags=re.M)))


def load():
    """dlopen the in-tree library and check it exports every declared symbol. No GPU needed."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = Path(os.environ.get("TBX_HIP_LIB", LIB_PATH))  # profiling builds only (tools/stage_clock.py)
    if not lib_path.exists():
        raise ImportError(
            f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(str(lib_path))
    for s in declared_symbols():
        if not hasattr(lib, s):
            raise ImportError(f"libtbx_hip.so does not export {s}")
    lib.tbx_error_string.restype = C.c_char_p
    lib.tbx_version.restype = C.c_int
    i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p
    lib.tbx_knn_embed.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.tbx_pose_embed.argtypes = [vp, i64, vp, vp, i32, vp, i32, i32, vp]
    lib.tbx_rel_pose_dense.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.tbx_diffbar_reward.argtypes = [vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, vp, vp, vp]
    lib.tbx_knarpe_attn_fwd.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp, vp]
    lib.tbx_knarpe_attn_fwd_folded.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp, vp, vp]
    lib.tbx_knarpe_attn_fwd_mfma.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp, vp]
    lib.tbx_knarpe_attn_fwd_mfma_dropout_tb.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp, f32, vp, C.c_uint32,
                                                        i32, i32, vp]
    lib.tbx_knarpe_dec_mid.argtypes = [C.POINTER(DecMid), vp]
    lib.tbx_knarpe_dec_layer.argtypes = [C.POINTER(DecLayer), vp]
    lib.tbx_knn_embed_multi.argtypes = [C.POINTER(KnnJob), i32, vp, vp, i32, vp]
    lib.tbx_knn_embed_multi_pe.argtypes = [C.POINTER(KnnJob), i32, vp, vp, i32, C.POINTER(PoseEmbedJob), vp]
    lib.tbx_knarpe_attn_bwd.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp,
                                        C.POINTER(C.c_void_p), vp, vp, vp, vp]
    lib.tbx_knarpe_attn_fwd_dropout.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp,
                                                f32, vp, C.c_uint32, vp]
    lib.tbx_knarpe_attn_bwd_dropout.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp,
                                                C.POINTER(C.c_void_p), vp, vp, vp, f32, vp, C.c_uint32, vp]
    lib.tbx_knarpe_attn_fwd_dropout_tb.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp, vp, vp,
                                                   f32, vp, C.c_uint32, i32, i32, vp]
    lib.tbx_knarpe_attn_bwd_dropout_tb.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp,
                                                   C.POINTER(C.c_void_p), vp, vp, vp, f32, vp, C.c_uint32, i32, i32, vp]
    lib.tbx_knarpe_attn_bwd_gather_tb.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp,
                                                  C.POINTER(C.c_void_p), vp, vp, vp, f32, vp, C.c_uint32, i32, i32,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), vp, vp]
    lib.tbx_keyed_dropout.argtypes = [vp, vp, i64, i32, i32, f32, vp, C.c_uint32, i32, i32, vp]
    lib.tbx_linear_wgrad_splits.argtypes = [i64, i32, i32]
    lib.tbx_linear_wgrad.argtypes = [vp, i32, vp, i32, i64, i32, i32, vp, vp, vp, i32, vp]
    lib.tbx_linear_wgrad_bf16.argtypes = lib.tbx_linear_wgrad.argtypes
    lib.tbx_residual_drop_fwd.argtypes = [vp, vp, vp, vp, i64, i32, C.c_float, vp, C.c_uint32, i32, i32, i32, vp, vp]
    lib.tbx_residual_drop_bwd.argtypes = [vp, vp, vp, i64, i32, C.c_float, vp, C.c_uint32, i32, i32, i32, vp, vp, vp]
    lib.tbx_relu_drop_fwd.argtypes = [vp, i64, i32, C.c_float, vp, C.c_uint32, i32, i32, i32, vp, vp]
    lib.tbx_relu_drop_bwd.argtypes = [vp, vp, i64, i32, C.c_float, vp, vp]
    lib.tbx_pair_bias_relu.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.tbx_pointnet_tail_fwd.argtypes = [vp, vp, i64, i32, i32, C.c_float, vp, C.c_uint32, i32, i32, i32, vp, vp]
    lib.tbx_pointnet_tail_bwd.argtypes = [vp, vp, vp, i64, i32, i32, C.c_float, vp, vp]
    lib.tbx_masked_maxpool_fwd.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    lib.tbx_masked_maxpool_bwd.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp]
    lib.tbx_layernorm_fwd.argtypes = [vp, vp, vp, C.c_float, i64, i32, vp, vp, vp, vp]
    lib.tbx_layernorm_bwd_partials.argtypes = [i64]
    lib.tbx_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp]
    lib.tbx_layernorm_bwd_add.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]
    lib.tbx_train_chain_fwd.argtypes = [C.POINTER(TrainChainArgs), vp, i64, i64, i32, i32, vp]
    lib.tbx_train_chain_fwd_windows.argtypes = [C.POINTER(TrainChainArgs), vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.tbx_train_chain_bwd.argtypes = [C.POINTER(TrainChainArgs), vp, i64, i64, vp, vp, vp]
    lib.tbx_knn_inverse.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.tbx_knarpe_attn_bwd_gather.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(AttnSeg), i32, vp, i32, vp,
                                               C.POINTER(C.c_void_p), vp, vp, vp, f32, vp, C.c_uint32, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), vp, vp]
    lib.tbx_pack_weight_size.argtypes = [i32, i32, i32]
    lib.tbx_pack_weight_size.restype = C.c_int64
    lib.tbx_pack_weight.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.tbx_pack_weight_split.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.tbx_pack_weight_gemv_size.argtypes = [i32, i32, i32]
    lib.tbx_pack_weight_gemv_size.restype = C.c_int64
    lib.tbx_pack_weight_gemv.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.tbx_layer_tile.argtypes = [C.POINTER(LayerTile), vp]
    lib.tbx_heads_tile.argtypes = [C.POINTER(HeadsTile), vp]
    lib.tbx_window_tile.argtypes = [C.POINTER(WindowTile), vp]
    lib.tbx_layer_tile_bf16.argtypes = [C.POINTER(LayerTile), vp]
    lib.tbx_heads_tile_bf16.argtypes = [C.POINTER(HeadsTile), vp]
    lib.tbx_window_tile_bf16.argtypes = [C.POINTER(WindowTile), vp]
    lib.tbx_front.argtypes = [C.POINTER(Front), vp]
    lib.tbx_front_pair.argtypes = [C.POINTER(Front), C.POINTER(Front), vp]
    lib.tbx_knarpe_dec_layer_pair.argtypes = [C.POINTER(DecLayer), C.POINTER(DecLayer), vp]
    lib.tbx_tall_linear.argtypes = [vp, C.c_int64, i32, i32, vp, i32, i32, i32, vp, i32, vp]
    lib.tbx_tall_linear_bf16.argtypes = lib.tbx_tall_linear.argtypes
    lib.tbx_tall_linear_dual.argtypes = [vp, C.c_int64, i32, i32, vp, i32, i32, i32, vp, i32, vp, i32, vp]
    lib.tbx_tall_linear_dual_bf16.argtypes = lib.tbx_tall_linear_dual.argtypes
    lib.tbx_tall_linear_relu_drop.argtypes = [vp, C.c_int64, i32, i32, vp, i32, i32, vp, i32, f32, vp, C.c_uint32, i32, i32, i32, vp]
    lib.tbx_tall_linear_relu_drop_bf16.argtypes = lib.tbx_tall_linear_relu_drop.argtypes
    lib.tbx_tl_tail_tile.argtypes = [vp, i64, C.POINTER(TlTail), vp]
    lib.tbx_tl_tail_tile_bf16.argtypes = lib.tbx_tl_tail_tile.argtypes
    lib.tbx_pack_weight_mfma32_size.argtypes = [i32, i32, i32]
    lib.tbx_pack_weight_mfma32_size.restype = C.c_int64
    lib.tbx_pack_weight_mfma32.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.tbx_pack_weight_mfma32_multi.argtypes = [C.POINTER(PackJob), i32, vp]
    lib.tbx_rowchain_live.argtypes = [C.POINTER(Stage), i32, i64, i32, i32, i32, i32, vp]
    lib.tbx_rowchain.argtypes = [C.POINTER(Stage), i32, i64, i32, i32, i32, vp]
    lib.tbx_rowchain_ex.argtypes = [C.POINTER(Stage), i32, i64, i32, i32, i32, i32, i32, vp]
    lib.tbx_agent_prep.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                   i32, vp, vp, vp]
    lib.tbx_tl_prep.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.tbx_map_prep.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.tbx_sim_step.argtypes = [C.POINTER(SimState), vp]
    lib.tbx_sim_step_parts.argtypes = [C.POINTER(SimState), i32, vp]
    lib.tbx_sim_step_tl_prep.argtypes = [C.POINTER(SimState), i32, vp, i32, vp, vp, vp]
    lib.tbx_rule_tables.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.tbx_rule_grid.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.tbx_rule_grid_cells.argtypes = []
    lib.tbx_rule_check.argtypes = [C.POINTER(RuleCtx), vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.tbx_rule_accumulate.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.tbx_filter_futures.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp]
    lib.tbx_attn_fold_fwd.argtypes = [vp] * 14
    lib.tbx_attn_fold_bwd.argtypes = [vp] * 19
    lib.tbx_rule_navi_check.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    for name in ("tbx_layer_tile", "tbx_heads_tile", "tbx_window_tile", "tbx_front", "tbx_tall_linear", "tbx_pack_weight_mfma32", "tbx_pack_weight_mfma32_multi", "tbx_pack_weight", "tbx_pack_weight_split", "tbx_pack_weight_gemv", "tbx_rowchain_live", "tbx_knarpe_attn_fwd_folded", "tbx_knarpe_dec_mid", "tbx_knarpe_dec_layer", "tbx_knn_embed_multi", "tbx_knn_embed_multi_pe", "tbx_knn_embed", "tbx_pose_embed", "tbx_knarpe_attn_fwd", "tbx_knarpe_attn_bwd", "tbx_knarpe_attn_fwd_dropout", "tbx_knarpe_attn_bwd_dropout", "tbx_knarpe_attn_bwd_gather", "tbx_knarpe_attn_fwd_dropout_tb", "tbx_knarpe_attn_bwd_dropout_tb", "tbx_knarpe_attn_bwd_gather_tb", "tbx_keyed_dropout", "tbx_linear_wgrad_splits", "tbx_linear_wgrad", "tbx_linear_wgrad_bf16", "tbx_tall_linear_bf16", "tbx_tl_tail_tile", "tbx_tl_tail_tile_bf16", "tbx_tall_linear_dual", "tbx_tall_linear_dual_bf16", "tbx_layernorm_fwd", "tbx_layernorm_bwd_partials", "tbx_layernorm_bwd", "tbx_layernorm_bwd_add", "tbx_residual_drop_fwd", "tbx_residual_drop_bwd", "tbx_relu_drop_fwd", "tbx_relu_drop_bwd", "tbx_pair_bias_relu", "tbx_pointnet_tail_fwd", "tbx_pointnet_tail_bwd", "tbx_masked_maxpool_fwd", "tbx_masked_maxpool_bwd", "tbx_train_chain_fwd", "tbx_train_chain_fwd_windows", "tbx_train_chain_bwd", "tbx_knn_inverse", "tbx_rowchain", "tbx_rowchain_ex", "tbx_agent_prep", "tbx_tl_prep",
                 "tbx_map_prep", "tbx_sim_step", "tbx_sim_step_parts", "tbx_sim_step_tl_prep", "tbx_rule_tables", "tbx_rule_grid", "tbx_rule_grid_cells", "tbx_rule_check", "tbx_rule_accumulate", "tbx_filter_futures", "tbx_rule_navi_check", "tbx_attn_fold_fwd", "tbx_attn_fold_bwd", "tbx_tall_linear_relu_drop", "tbx_tall_linear_relu_drop_bf16", "tbx_front_pair", "tbx_knarpe_dec_layer_pair",
                 "tbx_rel_pose_dense", "tbx_diffbar_reward", "tbx_knarpe_attn_fwd_mfma", "tbx_knarpe_attn_fwd_mfma_dropout_tb"):
        getattr(lib, name).restype = C.c_int
    if lib.tbx_version() != 4:
        raise ImportError("libtbx_hip.so ABI version mismatch")
    # (an entry point without argtypes would get 64-bit handles - stream pointers under graph capture - as C ints)
    untyped = [s for s in declared_symbols() if getattr(lib, s).argtypes is None and s not in ("tbx_error_string", "tbx_version")]
    if untyped:
        raise ImportError(f"abi.load(): no argtypes for {untyped}")
    _lib = lib
    return lib
