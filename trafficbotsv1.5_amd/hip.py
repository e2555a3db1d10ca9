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

from . import abi
from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import HEADER_PATH, LIB_PATH, load  # noqa: F401


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: tbx error {rc}: {load().tbx_error_string(rc).decode()}")


def _ptr(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tbx kernels need device tensors (HIP); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _cptr(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is not None and not t.is_contiguous():
        raise RuntimeError("tbx kernels need contiguous tensors")
    return _ptr(t, dtype)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


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


class Seg:
    """One target segment of a KNARPE attention call: a K/V table + the KNN set that indexes it."""

    def __init__(self, kv, k_off, v_off, n_tgt, idx, invalid, emb=None, batch_div=1, rel=None):
        """emb [n,S,k,128] (materialised embedding) or rel [n,S,k,3] (relative pose; embedding rebuilt in-kernel)."""
        assert kv.dim() == 2 and kv.stride(1) == 1 and kv.dtype in (torch.float32, torch.bfloat16)
        assert (emb is None) != (rel is None), "exactly one of emb / rel"
        self.kv, self.k_off, self.v_off, self.n_tgt, self.batch_div = kv, k_off, v_off, n_tgt, batch_div
        self.idx, self.invalid, self.emb, self.rel = idx, invalid, emb, rel
        self.k = idx.shape[-1]

    def c(self) -> AttnSeg:
        return AttnSeg(_ptr(self.kv), _cptr(self.idx, torch.int32), _cptr(self.invalid, torch.uint8),
                       _cptr(self.emb, torch.float32), _cptr(self.rel, torch.float32), self.kv.stride(0), self.k_off, self.v_off,
                       self.n_tgt, self.batch_div, self.k, int(self.kv.dtype == torch.bfloat16))


def _drop_args(drop):
    if drop is None:
        return 0.0, None, 0, 1, 0
    p, seed, call = drop[:3]
    tb, t0 = (drop[3], drop[4]) if len(drop) > 3 else (1, 0)
    return p, seed, call, int(tb), int(t0)


def keyed_dropout(x: torch.Tensor, p: float, seed: torch.Tensor, site: int, rows_per_scene: int, time_batch: int = 1,
                  time0: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tbx_keyed_dropout on x viewed as [rows, cols = x.shape[-1]] (contiguous); rows_per_scene = rows per batch entry."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols if cols else 0
    y = torch.empty_like(x) if out is None else out
    rc = load().tbx_keyed_dropout(_ptr(x), _ptr(y), rows, cols, rows_per_scene, float(p), _ptr(seed, torch.int64), int(site),
                                  int(time_batch), int(time0), stream_ptr())
    _check(rc, "tbx_keyed_dropout")
    return y


def tall_linear_ok(x: torch.Tensor, k: int, n: int) -> bool:
    """Shapes tbx_tall_linear takes: row-major fp32 rows of k values (a 2-D view after flattening the leading dimensions), k and n
    multiples of 128 up to 1024, 16-byte aligned, leading dimension a multiple of 4."""
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == k and k % 128 == 0 and n % 128 == 0 and k <= 1024 and n <= 1024
            and x.stride(-1) == 1 and x.data_ptr() % 16 == 0)


def tall_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, wt: bool = False, relu: bool = False,
                bf16: bool = False, out: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x W^T (+ b) over many rows on the split-bf16 matrix path (tbx_tall_linear; bf16: ONE bf16 product per term,
    tbx_tall_linear_bf16). wt: w is stored [k x n] (the input gradient dx = dy W of a Linear with weight W [n_out, n_in]: x = dy,
    w = W, wt = True)."""
    n, k = (w.shape[1], w.shape[0]) if wt else (w.shape[0], w.shape[1])
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    img = packed_weight(w, b, wt=wt, mfma32=True)
    y = torch.empty(x2.shape[0], n, dtype=torch.float32, device=x.device) if out is None else out
    assert y.shape == (x2.shape[0], n) and y.is_contiguous() and y.dtype == torch.float32
    if out16 is not None:  # the same rows as bfloat16 as well (tbx_tall_linear_dual)
        assert out16.shape == y.shape and out16.dtype == torch.bfloat16 and out16.is_contiguous()
        fn = load().tbx_tall_linear_dual_bf16 if bf16 else load().tbx_tall_linear_dual
        _check(fn(_ptr(x2, torch.float32), x2.shape[0], k, x2.stride(0), _ptr(img, torch.float32), n, int(b is not None), int(relu),
                  _ptr(y), n, _ptr(out16, torch.bfloat16), n, stream_ptr()), "tbx_tall_linear_dual")
        return y.view(*x.shape[:-1], n)
    fn = load().tbx_tall_linear_bf16 if bf16 else load().tbx_tall_linear
    _check(fn(_ptr(x2, torch.float32), x2.shape[0], k, x2.stride(0), _ptr(img, torch.float32), n, int(b is not None), int(relu),
              _ptr(y), n, stream_ptr()), "tbx_tall_linear")
    return y.view(*x.shape[:-1], n)


def linear_wgrad_ok(dy: torch.Tensor, x: torch.Tensor) -> bool:
    """Shapes tbx_linear_wgrad takes: 2-D row-major fp32 views, n, k and both leading dimensions multiples of 4, 16-B aligned."""
    return (dy.dim() == 2 and x.dim() == 2 and dy.is_cuda and dy.dtype == torch.float32 and x.dtype == torch.float32
            and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[1] % 4 == 0 and x.shape[1] % 4 == 0
            and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, want_db: bool = True, bf16: bool = False):
    """(dw [n,k], db [n] | None) = (dy^T x, sum_rows dy) for dy [rows,n], x [rows,k] (tbx_linear_wgrad; bf16: one bf16 product per
    term with fp32 accumulation, tbx_linear_wgrad_bf16)."""
    rows, n = dy.shape
    k = x.shape[1]
    lib = load()
    splits = lib.tbx_linear_wgrad_splits(rows, n, k)
    if splits <= 0:
        _check(int(splits), "tbx_linear_wgrad_splits")
    scratch = torch.empty(splits, n * k + n, dtype=torch.float32, device=dy.device)
    dw = torch.empty(n, k, dtype=torch.float32, device=dy.device)
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if want_db else None
    fn = lib.tbx_linear_wgrad_bf16 if bf16 else lib.tbx_linear_wgrad
    _check(fn(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), rows, n, k, _ptr(dw), _ptr(db), _ptr(scratch), splits, stream_ptr()), "tbx_linear_wgrad")
    return dw, db


NO_DROP = (0.0, None, 0, 1, 1, 0)  # (p, seed, site, rows_per_scene, time_batch, time0) of the glue ops without dropout


def glue_ok(x: torch.Tensor) -> bool:
    """Tensors the one-pass glue kernels (tbx_residual_drop_*, tbx_relu_drop_*) take: fp32 on the device, last dimension % 4 == 0."""
    return x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.shape[-1] % 4 == 0 and x.numel() > 0


def _drop6(drop):
    p, seed, site, rps, tb, t0 = drop
    return float(p), (_ptr(seed, torch.int64) if seed is not None else None), int(site), int(rps), int(tb), int(t0)


def residual_drop_fwd(x, y, zero_y, zero_out, drop=NO_DROP):
    """zero_out[row] ? 0 : x + dropout(zero_y[row] ? 0 : y); zero_* u8 per row or None."""
    assert glue_ok(x) and x.is_contiguous() and y.is_contiguous() and y.shape == x.shape and y.dtype == torch.float32
    cols = x.shape[-1]
    rows = x.numel() // cols
    for z in (zero_y, zero_out):
        assert z is None or (z.dtype == torch.uint8 and z.is_contiguous() and z.numel() == rows)
    out = torch.empty_like(x)
    p, seed, site, rps, tb, t0 = _drop6(drop)
    _check(load().tbx_residual_drop_fwd(_ptr(x), _ptr(y), _ptr(zero_y), _ptr(zero_out), rows, cols, p, seed, site, rps, tb, t0, _ptr(out),
                                        stream_ptr()), "tbx_residual_drop_fwd")
    return out


def residual_drop_bwd(dout, zero_y, zero_out, drop=NO_DROP):
    """-> (dy, dx); dx is dout itself when there is no zero_out mask."""
    assert glue_ok(dout) and dout.is_contiguous()
    cols = dout.shape[-1]
    rows = dout.numel() // cols
    dy = torch.empty_like(dout)
    dx = torch.empty_like(dout) if zero_out is not None else None
    p, seed, site, rps, tb, t0 = _drop6(drop)
    _check(load().tbx_residual_drop_bwd(_ptr(dout), _ptr(zero_y), _ptr(zero_out), rows, cols, p, seed, site, rps, tb, t0, _ptr(dy), _ptr(dx),
                                        stream_ptr()), "tbx_residual_drop_bwd")
    return dy, (dx if dx is not None else dout)


def relu_drop_fwd(z, drop=NO_DROP):
    assert glue_ok(z) and z.is_contiguous()
    cols = z.shape[-1]
    h = torch.empty_like(z)
    p, seed, site, rps, tb, t0 = _drop6(drop)
    _check(load().tbx_relu_drop_fwd(_ptr(z), z.numel() // cols, cols, p, seed, site, rps, tb, t0, _ptr(h), stream_ptr()), "tbx_relu_drop_fwd")
    return h


def relu_drop_bwd(dh, h, p: float):
    assert glue_ok(dh) and dh.is_contiguous() and h.is_contiguous() and h.shape == dh.shape
    cols = dh.shape[-1]
    dz = torch.empty_like(dh)
    _check(load().tbx_relu_drop_bwd(_ptr(dh), _ptr(h), dh.numel() // cols, cols, float(p), _ptr(dz), stream_ptr()), "tbx_relu_drop_bwd")
    return dz


def pointnet_tail_ok(z: torch.Tensor) -> bool:
    """z [G, W, 64] fp32 on the device, W <= 16: the shapes tbx_pointnet_tail_* / tbx_masked_maxpool_* take."""
    return z.is_cuda and z.dtype == torch.float32 and z.dim() == 3 and z.shape[2] == 64 and 0 < z.shape[1] <= 16 and z.shape[0] > 0


def pointnet_tail_fwd(z: torch.Tensor, invalid_u8: torch.Tensor, drop=None) -> torch.Tensor:
    """[relu(z) (* keyed dropout) | its max over the group's valid rows], invalid rows zeroed. drop = None or (p, seed int64[1] device
    tensor, site, rows_per_scene, time_batch, time0) - tbx_keyed_dropout's arguments for the [G * W, 64] view."""
    assert pointnet_tail_ok(z) and z.is_contiguous() and invalid_u8.dtype == torch.uint8 and invalid_u8.is_contiguous()
    G, W, Cc = z.shape
    assert invalid_u8.numel() == G * W
    out = torch.empty(G, W, 2 * Cc, dtype=torch.float32, device=z.device)
    p, seed, site, rps, tb, t0 = drop if drop is not None else (0.0, None, 0, 1, 1, 0)
    _check(load().tbx_pointnet_tail_fwd(_ptr(z), _ptr(invalid_u8), G, W, Cc, float(p), _ptr(seed, torch.int64) if seed is not None else None,
                                        int(site), int(rps), int(tb), int(t0), _ptr(out), stream_ptr()), "tbx_pointnet_tail_fwd")
    return out


def pointnet_tail_bwd(dout: torch.Tensor, out: torch.Tensor, invalid_u8: torch.Tensor, p: float) -> torch.Tensor:
    G, W, C2 = out.shape
    assert dout.shape == out.shape and dout.is_contiguous() and dout.dtype == torch.float32
    dz = torch.empty(G, W, C2 // 2, dtype=torch.float32, device=out.device)
    _check(load().tbx_pointnet_tail_bwd(_ptr(dout), _ptr(out), _ptr(invalid_u8), G, W, C2 // 2, float(p), _ptr(dz), stream_ptr()),
           "tbx_pointnet_tail_bwd")
    return dz


def masked_maxpool_fwd(x: torch.Tensor, invalid_u8: torch.Tensor) -> torch.Tensor:
    G, W, C2 = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32 and invalid_u8.numel() == G * W
    y = torch.empty(G, C2, dtype=torch.float32, device=x.device)
    _check(load().tbx_masked_maxpool_fwd(_ptr(x), _ptr(invalid_u8), G, W, C2, _ptr(y), stream_ptr()), "tbx_masked_maxpool_fwd")
    return y


def masked_maxpool_bwd(dy: torch.Tensor, x: torch.Tensor, invalid_u8: torch.Tensor) -> torch.Tensor:
    G, W, C2 = x.shape
    assert dy.is_contiguous() and dy.shape == (G, C2) and dy.dtype == torch.float32
    dx = torch.empty_like(x)
    _check(load().tbx_masked_maxpool_bwd(_ptr(dy), _ptr(x), _ptr(invalid_u8), G, W, C2, _ptr(dx), stream_ptr()), "tbx_masked_maxpool_bwd")
    return dx


def layernorm_bwd_ok(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 128 and x.numel() > 0


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """(y, mean [rows], rstd [rows]) of LayerNorm_128 (tbx_layernorm_fwd)."""
    assert layernorm_bwd_ok(x) and x.is_contiguous()
    rows = x.numel() // 128
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(load().tbx_layernorm_fwd(_ptr(x), _ptr(gamma.contiguous(), torch.float32), _ptr(beta.contiguous(), torch.float32), float(eps), rows, 128,
                                    _ptr(y), _ptr(mean), _ptr(rstd), stream_ptr()), "tbx_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
    """(dx, dgamma, dbeta) of y = LayerNorm_128(x) * gamma + beta given dy, with the forward's per-row mean / rstd (tbx_layernorm_bwd)."""
    assert layernorm_bwd_ok(x) and x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and dy.dtype == torch.float32
    rows = x.numel() // 128
    assert mean.numel() == rows and rstd.numel() == rows and mean.is_contiguous() and rstd.is_contiguous()
    lib = load()
    n = lib.tbx_layernorm_bwd_partials(rows)
    scratch = torch.empty(n, 256, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dg = torch.empty(128, dtype=torch.float32, device=x.device)
    db = torch.empty(128, dtype=torch.float32, device=x.device)
    _check(lib.tbx_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma.contiguous(), torch.float32), _ptr(mean, torch.float32), _ptr(rstd, torch.float32),
                                 rows, 128, _ptr(dx), _ptr(dg), _ptr(db), _ptr(scratch), stream_ptr()), "tbx_layernorm_bwd")
    return dx, dg, db


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


def knarpe_attn_bwd(qbuf, q_off: int, qt_off: int, rpe_k_bias, n_batch: int, n_src: int, segs: Sequence[Seg], dout, dqbuf,
                    dkv: Sequence[torch.Tensor], dbias_k, freqs_xy=None, freqs_yaw=None, drop=None):
    arr = (AttnSeg * len(segs))(*[s.c() for s in segs])
    dk = (C.c_void_p * len(segs))(*[_ptr(t, torch.float32) for t in dkv])
    p, seed, call, tb, t0 = _drop_args(drop)
    rc = load().tbx_knarpe_attn_bwd_dropout_tb(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, _ptr(rpe_k_bias, torch.float32),
                                            n_batch, n_src, arr, len(segs), _ptr(dout, torch.float32), dout.stride(0),
                                            _ptr(dqbuf, torch.float32), dk, _ptr(dbias_k, torch.float32), _cptr(freqs_xy),
                                            _cptr(freqs_yaw), float(p), _ptr(seed, torch.int64), int(call), tb, t0, stream_ptr())
    _check(rc, "tbx_knarpe_attn_bwd")


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
    _check(load().tbx_front(C.byref(f), stream_ptr()), "tbx_front")
    return outs


def padded_weight(w: torch.Tensor, k_pad: int) -> torch.Tensor:
    """w [n, k] zero-padded to k_pad columns; cached like packed_weight (per parameter version, or per training step in PACK_SCOPE)."""
    if w.shape[1] == k_pad:
        return w
    key = ("padded", id(w), k_pad)
    stamp = (w._version, w.data_ptr())
    cache = PACK_SCOPE if PACK_SCOPE is not None else w.__dict__.setdefault("_tbx_padded", {})
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        out = torch.zeros(w.shape[0], k_pad, dtype=torch.float32, device=w.device)
        out[:, :w.shape[1]].copy_(w)
    cache[key] = (stamp, out)
    if PACK_SCOPE is not None:
        PACK_SCOPE.setdefault("_keep", {})[id(w)] = w
    return out


def knn_inverse(idx, invalid, n_tgt: int, tgt_batch_div: int = 1):
    """Inverse lists of a K-nearest set idx / invalid [n_batch, n_src, k] -> (inv_ptr [n_tables, n_tgt+1], inv_list [n_tables, cap])."""
    n, S, k = idx.shape
    nt = n // tgt_batch_div
    ptr = torch.empty(nt, n_tgt + 1, dtype=torch.int32, device=idx.device)
    lst = torch.empty(nt, S * tgt_batch_div * k, dtype=torch.int32, device=idx.device)
    rc = load().tbx_knn_inverse(_cptr(idx, torch.int32), _cptr(invalid, torch.uint8), n, S, k, n_tgt, tgt_batch_div, _ptr(ptr), _ptr(lst),
                                stream_ptr())
    _check(rc, "tbx_knn_inverse")
    return ptr, lst


def knarpe_attn_bwd_gather(qbuf, q_off: int, qt_off: int, rpe_k_bias, n_batch: int, n_src: int, segs: Sequence[Seg], dout, dqbuf,
                           dkv: Sequence[torch.Tensor], dbias_rows, inv: Sequence, freqs_xy=None, freqs_yaw=None, drop=None):
    """Backward through inverse K-nearest lists (inv[i] = knn_inverse(...) of segment i): no dK / dV atomics."""
    arr = (AttnSeg * len(segs))(*[s.c() for s in segs])
    dk = (C.c_void_p * len(segs))(*[_ptr(t, torch.float32) for t in dkv])
    ip = (C.c_void_p * len(segs))(*[_cptr(p, torch.int32) for p, _ in inv])
    il = (C.c_void_p * len(segs))(*[_cptr(l, torch.int32) for _, l in inv])
    coef = torch.empty(n_batch * n_src, sum(s.k for s in segs), 8, dtype=torch.float32, device=qbuf.device)
    p, seed, call, tb, t0 = _drop_args(drop)
    rc = load().tbx_knarpe_attn_bwd_gather_tb(_ptr(qbuf, torch.float32), qbuf.stride(0), q_off, qt_off, _ptr(rpe_k_bias, torch.float32),
                                           n_batch, n_src, arr, len(segs), _ptr(dout, torch.float32), dout.stride(0),
                                           _ptr(dqbuf, torch.float32), dk, _ptr(dbias_rows, torch.float32), _cptr(freqs_xy),
                                           _cptr(freqs_yaw), float(p), _ptr(seed, torch.int64), int(call), tb, t0, ip, il, _ptr(coef),
                                           stream_ptr())
    _check(rc, "tbx_knarpe_attn_bwd_gather")


def dropout_keep_mask(seed: int, call: int, n_rows: int, k_tot: int, p: float, n_head: int = 4, step: int = 0) -> torch.Tensor:
    """Host restatement of the kernels' counter-based mask (csrc/attn.hip DropKey): bool [n_rows, n_head, k_tot] - for tests
    and for anyone who needs the mask a (seed, call) pair produces."""
    import numpy as np

    sd = np.uint64(seed % (1 << 64))
    m32 = np.uint64(0xFFFFFFFF)
    lo = np.uint32(sd & m32) ^ np.uint32((call * 0x85EBCA6B) & 0xFFFFFFFF) ^ np.uint32((step * 0x27D4EB2F) & 0xFFFFFFFF)
    hi = np.uint32((int((sd >> np.uint64(32)) & m32) + call * 0xC2B2AE35 + step * 0x165667B1) & 0xFFFFFFFF)
    row = np.arange(n_rows, dtype=np.uint32)[:, None, None]
    h = np.arange(n_head, dtype=np.uint32)[None, :, None]
    t = np.arange(k_tot, dtype=np.uint32)[None, None, :]
    with np.errstate(over="ignore"):
        x = ((row * np.uint32(128) + t) * np.uint32(4) + h) ^ lo
        x = x * np.uint32(0x9E3779B1)
        x = x ^ hi
        x = x ^ (x >> np.uint32(16))
        x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> np.uint32(15))
        x = x * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    th = p * 4294967296.0
    th = np.uint32(1 if 0 < th < 1 else int(th))
    return torch.from_numpy(x >= th)


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


def train_chain_fwd(args: TrainChainArgs, mean: torch.Tensor, stride_n: int, stride_t: int, t0: int, t1: int):
    _check(load().tbx_train_chain_fwd(C.byref(args), _ptr(mean, torch.float32), stride_n, stride_t, t0, t1, stream_ptr()), "tbx_train_chain_fwd")


def train_chain_fwd_windows(args: TrainChainArgs, mean: Optional[torch.Tensor], stride_n: int, stride_t: int, t0: int, t1: int, hv, hp, hm, valid, navi_valid):
    """tbx_train_chain_fwd over [t0, t1) (t0 == t1: none) + the policy inputs of step t1 + 1 into hv u8 [n,A,W], hp / hm f32 [n,A,W,3], valid /
    navi_valid [n,A] (bool or u8 storage)."""
    _check(load().tbx_train_chain_fwd_windows(C.byref(args), _cptr(mean, torch.float32), stride_n, stride_t, t0, t1, _ptr(hv, torch.uint8),
                                              _ptr(hp, torch.float32), _ptr(hm, torch.float32), _ptr(valid), _ptr(navi_valid), stream_ptr()),
           "tbx_train_chain_fwd_windows")


def train_chain_bwd(args: TrainChainArgs, mean: torch.Tensor, stride_n: int, stride_t: int, d_reward: torch.Tensor, d_mean: torch.Tensor):
    _check(load().tbx_train_chain_bwd(C.byref(args), _ptr(mean, torch.float32), stride_n, stride_t, _ptr(d_reward, torch.float32),
                                      _ptr(d_mean, torch.float32), stream_ptr()), "tbx_train_chain_bwd")


def sim_step(state: SimState, parts: int = SIM_AGENTS | SIM_LIGHTS | SIM_ADVANCE, tl_prep=None):
    """tl_prep = (tl_invalid u8 [n*L], attr f32 [n*L*W, ld], row_invalid u8 [n*L*W]): the lights' update also writes the tbx_tl_prep
    rows of their new windows (tbx_sim_step_tl_prep)."""
    if tl_prep is not None:
        inv, attr, row_inv = tl_prep
        _check(load().tbx_sim_step_tl_prep(C.byref(state), parts, _cptr(inv, torch.uint8), attr.stride(0), _ptr(attr, torch.float32),
                                           _ptr(row_inv, torch.uint8), stream_ptr()), "tbx_sim_step_tl_prep")
        return
    _check(load().tbx_sim_step_parts(C.byref(state), parts, stream_ptr()), "tbx_sim_step_parts")


# ------------------------------------------------------------------------------------------------ rowchain builder
def rule_tables(mp_valid_u8, mp_type_idx_u8, mp_pos, mp_dir):
    """-> (seg [n,M*N,4], n_seg [n] i32, lane [n,M*N,2], n_lane [n] i32): compacted road-edge segments / lane-centre nodes."""
    n, M, N = mp_valid_u8.shape
    dev = mp_pos.device
    seg = torch.empty(n, M * N, 4, dtype=torch.float32, device=dev)
    lane = torch.empty(n, M * N, 2, dtype=torch.float32, device=dev)
    n_seg = torch.empty(n, dtype=torch.int32, device=dev)
    n_lane = torch.empty(n, dtype=torch.int32, device=dev)
    rc = load().tbx_rule_tables(_cptr(mp_valid_u8, torch.uint8), _cptr(mp_type_idx_u8, torch.uint8), _cptr(mp_pos, torch.float32),
                                _cptr(mp_dir, torch.float32), mp_pos.shape[-1], n, M, N, _ptr(seg), _ptr(n_seg), _ptr(lane),
                                _ptr(n_lane), stream_ptr())
    _check(rc, "tbx_rule_tables")
    return seg, n_seg, lane, n_lane


def rule_check(ctx: RuleCtx, valid_u8, pose, motion, tl_state_u8, ld_t: int, t0: int, n_t: int, flags):
    rc = load().tbx_rule_check(C.byref(ctx), _cptr(valid_u8, torch.uint8), _cptr(pose, torch.float32), _cptr(motion, torch.float32),
                               _cptr(tl_state_u8, torch.uint8), ld_t, t0, n_t, _cptr(flags, torch.uint8), stream_ptr())
    _check(rc, "tbx_rule_check")


def rule_accumulate(raw, n_rows: int, ld_t: int, t0: int, n_t: int, acc_state, passive_counter, out_now, out_acc):
    rc = load().tbx_rule_accumulate(_cptr(raw, torch.uint8), n_rows, ld_t, t0, n_t, _cptr(acc_state, torch.uint8),
                                    _cptr(passive_counter, torch.float32), _cptr(out_now, torch.uint8), _cptr(out_acc, torch.uint8),
                                    stream_ptr())
    _check(rc, "tbx_rule_accumulate")


def filter_futures(flags, col_bit: int, ag_role_any, n_scene: int, n_k: int, t_start: int, w_road_edge: float, n_keep: int,
                   pred_pose=None):
    """flags [n_scene*n_k, A, T] u8 bits, ag_role_any [n_scene, A] u8 -> (score [n_scene,n_k], idx [n_scene,n_keep] i32,
    trajs [n_scene, n_keep, A, T - t_start, 3] or None)."""
    A, T = flags.shape[-2:]
    dev = flags.device
    score = torch.empty(n_scene, n_k, dtype=torch.float32, device=dev)
    idx = torch.empty(n_scene, n_keep, dtype=torch.int32, device=dev)
    trajs = None if pred_pose is None else torch.empty(n_scene, n_keep, A, T - t_start, 3, dtype=torch.float32, device=dev)
    rc = load().tbx_filter_futures(_cptr(flags, torch.uint8), col_bit, _cptr(ag_role_any, torch.uint8), n_scene, n_k, A, T, t_start,
                                   w_road_edge, n_keep, _ptr(score), _ptr(idx), _cptr(pred_pose, torch.float32), _ptr(trajs),
                                   stream_ptr())
    _check(rc, "tbx_filter_futures")
    return score, idx, trajs


# Set to a dict for the duration of a training step (train_graph.training_step): chain kernels of the step's no-grad stepping
# pass then pack each weight ONCE PER STEP into this scope instead of the per-parameter cache. A captured training step
# (GraphedTrainStep) replays after the optimizer has moved the weights: a cached image from before the capture would be read by
# the replay without ever being re-packed (its tbx_pack_weight launch is not in the graph) - with the scope the packing is.
PACK_SCOPE: Optional[dict] = None


def packed_weight(w: torch.Tensor, bias: Optional[torch.Tensor] = None, wt: bool = False, groups: int = 1,
                  split: bool = False, gemv: bool = False, mfma32: bool = False) -> torch.Tensor:
    """tbx_pack_weight image of a LINEAR weight (+ bias). Cached on the weight's base tensor object (the nn.Parameter)
    per view and version of both tensors: re-packed after an in-place update (optimizer step, load_state_dict), reused
    otherwise - chains are rebuilt every eager step - and dropped with the parameter.
    split=True: the tbx_pack_weight_split image (bf16 hi + lo halves) for stages flagged F_WSPLIT.
    gemv=True: the tbx_pack_weight_gemv image (column streams) for the F_WGEMV stages of live-row chains.
    mfma32=True: the tbx_pack_weight_mfma32 image (per-wave units of bf16 hi + lo fragments) for tbx_layer_tile."""
    assert w.is_cuda and w.dim() == 2 and w.stride(1) == 1 and w.dtype == torch.float32
    base = w._base if w._base is not None else w
    bkey = None if bias is None else (bias.data_ptr(), bias.shape[0])
    key = (w.storage_offset(), tuple(w.shape), w.stride(0), wt, groups, bkey, split, gemv, mfma32)
    if PACK_SCOPE is not None:  # a training step: images live (and are re-packed) per step, see PACK_SCOPE
        cache, key = PACK_SCOPE, (id(base),) + key
        PACK_SCOPE.setdefault("_keep", {})[id(base)] = base  # ids stay unique while the scope lives
    else:
        cache = base.__dict__.setdefault("_tbx_packed", {})
    stamp = (w._version, w.data_ptr(), None if bias is None else bias._version)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    n, k = (w.shape[1], w.shape[0] // groups) if wt else (w.shape[0] // groups, w.shape[1])
    if bias is not None:
        assert bias.is_cuda and bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == groups * n
    lib = load()
    size = (lib.tbx_pack_weight_mfma32_size if mfma32 else (lib.tbx_pack_weight_gemv_size if gemv else lib.tbx_pack_weight_size))(n, k, groups)
    if size <= 0:
        _check(int(size), "tbx_pack_weight_size")
    out = torch.empty(size, dtype=torch.float32, device=w.device)
    fn = lib.tbx_pack_weight_mfma32 if mfma32 else (lib.tbx_pack_weight_gemv if gemv else (lib.tbx_pack_weight_split if split else lib.tbx_pack_weight))
    _check(fn(_ptr(w), _ptr(bias), n, k, w.stride(0), groups, int(wt), _ptr(out), stream_ptr()), "tbx_pack_weight")
    cache[key] = (stamp, out)
    return out


def stacked_linear(linears, pad_out_to: int = 0):
    """(W [G * n, k], b [G * n]) = the weights / biases of G equally shaped nn.Linear layers stacked along the output dimension
    (each block zero-padded to pad_out_to output rows if given): branches that read the same input become ONE LINEAR stage
    (G * n outputs), parallel branches one block-diagonal stage (groups = G). Cached like packed_weight: per parameter version,
    or per training step inside PACK_SCOPE."""
    ws, bs = [l.weight for l in linears], [l.bias for l in linears]
    key = ("stacked", tuple(id(w) for w in ws), pad_out_to)
    stamp = tuple((w._version, w.data_ptr(), b._version) for w, b in zip(ws, bs))
    cache = PACK_SCOPE if PACK_SCOPE is not None else ws[0].__dict__.setdefault("_tbx_stacked", {})
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    n, k = ws[0].shape
    npad = max(n, pad_out_to)
    with torch.no_grad():
        W = torch.zeros(len(ws) * npad, k, dtype=torch.float32, device=ws[0].device)
        B = torch.zeros(len(ws) * npad, dtype=torch.float32, device=ws[0].device)
        for g, (w, b) in enumerate(zip(ws, bs)):
            W[g * npad:g * npad + n].copy_(w)
            B[g * npad:g * npad + n].copy_(b)
    cache[key] = (stamp, W, B)
    if PACK_SCOPE is not None:
        PACK_SCOPE.setdefault("_keep", {})[id(ws[0])] = ws[0]
    return W, B


def group_tile_rows(group_rows: int, n_groups: int) -> int:
    """Tile height of a grouped chain. Small grids keep one group per 16-row tile (more workgroups, shortest critical path). Once
    the groups outnumber the CUs several times the per-stage fixed costs of a workgroup are worth sharing: the tile (32 or 48 rows)
    that wastes the fewest rows holds floor(tile / W) whole groups - 11-step windows: 4 in 48 rows (92 % of the MFMA rows used; 2 in
    32 rows: 69 %). Measured at 4096 windows of 11 rows: 415 us (16) -> ~230 us (32) -> see DESIGN.md (48). TBX_TILE48=0: never 48."""
    if n_groups < 1024 or group_rows > 24:
        return 32 if group_rows > 16 else 16
    use = lambda t: (t // group_rows) * group_rows / t
    cands = [t for t in ((32, 48) if os.environ.get("TBX_TILE48", "1") != "0" else (32,)) if t // group_rows >= 1]
    best = max(cands, key=lambda t: (use(t), -t))
    return best if best // group_rows >= 2 or group_rows > 16 else 16


class Chain:
    """Builds one tbx_rowchain program. Tensors handed to stages are kept alive by the chain; the encoded program
    holds raw device pointers, so a chain is valid as long as those tensors are not re-allocated."""

    def __init__(self, tile_rows: int = 16, ldw: int = 132, ldw1: Optional[int] = None, ld_aux: Optional[int] = None,
                 live_rows: int = 0):
        """ldw = LDS row width of BUF0 (floats); ldw1 / ld_aux default to ldw / 260 (tbx_rowchain), else tbx_rowchain_ex.
        live_rows in (1, 2, 4): a tbx_rowchain_live program - tiles of that many rows, LINEAR stages on the thread-per-column path
        (same results bit for bit; for launches of a few hundred rows at most)."""
        self.tile_rows, self.ldw, self.live_rows = tile_rows, ldw, live_rows
        assert live_rows in (0, 1, 2, 4) and (live_rows == 0 or tile_rows == 16)
        self.ldw1 = ldw if ldw1 is None else ldw1
        self.ld_aux = AUX_LD if ld_aux is None else ld_aux
        self.stages: List[Stage] = []
        self._keep = []
        self._arr = None
        self.pack_weights = Chain.pack_default
        self.split_bf16 = Chain.split_default

    pack_default = True  # LINEAR weights are handed to the kernel as tbx_pack_weight images (row-major kept for tests)
    # packed LINEAR stages on the three-product split-bf16 MFMA path (~1e-5 relative instead of exact fp32; include/tbx_hip.h
    # TBX_F_WSPLIT). Off by default: the exact-fp32 MFMA is the parity path; TBX_SPLIT_BF16=1 turns it on for a process.
    split_default = os.environ.get("TBX_SPLIT_BF16", "0") == "1"

    def _add(self, **kw):
        p0, p1, p2 = kw.pop("p0", None), kw.pop("p1", None), kw.pop("p2", None)
        for t in (p0, p1, p2):
            if t is not None:
                self._keep.append(t)
        st = Stage(**kw)
        st.p0, st.p1, st.p2 = _ptr(p0), _ptr(p1), _ptr(p2)
        self.stages.append(st)
        self._arr = None
        return self

    @staticmethod
    def _rows2d(t):
        assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view expected"
        return t

    def load(self, src, dst, dst_col=0, n=None, pad_to=0, accum=False, row_div=0, row_mod=0, row_idx=None, batch_mod=None):
        """dst[:, dst_col:+n] (=|+=) src[row_of(g), :n]. batch_mod=(rows_per_batch_here, rows_per_batch_src)."""
        n = src.shape[1] if n is None else n
        flags, div, k, p1 = (F_ACCUM if accum else 0), 0, pad_to, None
        if row_div:
            flags, div = flags | F_ROW_DIV, row_div
        elif row_mod:
            flags, div = flags | F_ROW_MOD, row_mod
        elif row_idx is not None:
            flags, p1 = flags | F_ROW_IDX, row_idx
        elif batch_mod is not None:
            flags, k, div = flags | F_ROW_BATCH_MOD, batch_mod[0], batch_mod[1]
        return self._add(op=OP_LOAD, dst=dst, dst_col=dst_col, n=n, k=k, flags=flags, div=div, ld=self._rows2d(src).stride(0),
                         p0=src, p1=p1)

    def load2(self, src, dst, dst_col, src_b, dst_b, dst_b_col):
        """Two row loads in ONE stage (one memory round trip): dst[:, dst_col:+src.shape[1]] = src and dst_b[:, dst_b_col:+..] =
        src_b (whole float4 rows on both sides, src_b at most 256 floats wide)."""
        a, b = self._rows2d(src), self._rows2d(src_b)
        return self._add(op=OP_LOAD, dst=dst, dst_col=dst_col, n=a.shape[1], k=0, flags=F_LOAD2, ld=a.stride(0), p0=a,
                         src=dst_b, src_col=dst_b_col, reserved=b.shape[1], ld2=b.stride(0), p2=b)

    def zero(self, dst, dst_col, n):
        return self._add(op=OP_LOAD, dst=dst, dst_col=dst_col, n=n, k=0, ld=1)

    def linear(self, src, src_col, dst, dst_col, weight, bias=None, relu=False, accum=False, wt=False, groups=1,
               src_stride=0, dst_stride=0, out=None, skip_rows=None, skip_is_valid=False, zero_skipped=False):
        """dst[:, dst_col:+n] (=|+=) act(src[:, src_col:+k] @ W^T + b), W = weight [n,k] (or [k,n] if wt).
        groups > 1: block-diagonal; weight holds the groups' blocks stacked along dim 0, group g reads
        src_col + g*src_stride and writes dst_col + g*dst_stride.
        dst = GLOBAL with out = [rows, ld] tensor: the result goes straight to out[g, dst_col:+n] (no LDS staging).
        skip_rows (packed weights only): u8 per global row; flagged rows (un-flagged with skip_is_valid) keep dst's old content
        (zero_skipped: are written as 0 instead: LINEAR + ROWMASK in one stage) -
        with accum into the residual buffer: x += flagged ? 0 : linear(...) in one stage."""
        w = self._rows2d(weight)
        n, k = (w.shape[1], w.shape[0] // groups) if wt else (w.shape[0] // groups, w.shape[1])
        flags = (F_ACCUM if accum else 0) | (F_WT if wt else 0)
        assert (dst == GLOBAL) == (out is not None)
        if out is not None and out.dtype == torch.bfloat16:  # a bf16 K/V table: rounded on the way out (TBX_F_OUT_BF16)
            assert self.live_rows or self.pack_weights
            flags |= F_OUT_BF16
        if self.live_rows:
            flags = (flags & ~F_WT) | F_WGEMV
            if skip_rows is not None:
                flags |= F_ROWSKIP | (F_MASK_INV if skip_is_valid else 0) | (F_ROWZERO if zero_skipped else 0)
            return self._add(op=OP_LINEAR, src=src, dst=dst, src_col=src_col, dst_col=dst_col, k=k, n=n,
                             act=ACT_RELU if relu else ACT_NONE, flags=flags, ld=k, p0=packed_weight(w, bias, wt, groups, gemv=True),
                             p1=skip_rows, p2=out, ld2=0 if out is None else self._rows2d(out).stride(0),
                             reserved=groups if groups > 1 else 0, div=(src_stride << 16) | dst_stride)
        if self.pack_weights:
            w, flags = packed_weight(w, bias, wt, groups, self.split_bf16), (flags & ~F_WT) | F_WPACK
            if self.split_bf16:
                flags |= F_WSPLIT
            if skip_rows is not None:
                flags |= F_ROWSKIP | (F_MASK_INV if skip_is_valid else 0) | (F_ROWZERO if zero_skipped else 0)
            return self._add(op=OP_LINEAR, src=src, dst=dst, src_col=src_col, dst_col=dst_col, k=k, n=n,
                             act=ACT_RELU if relu else ACT_NONE, flags=flags, ld=k, p0=w, p1=skip_rows, p2=out,
                             ld2=0 if out is None else self._rows2d(out).stride(0),
                             reserved=groups if groups > 1 else 0, div=(src_stride << 16) | dst_stride)
        assert skip_rows is None, "skip_rows needs packed weights"
        return self._add(op=OP_LINEAR, src=src, dst=dst, src_col=src_col, dst_col=dst_col, k=k, n=n,
                         act=ACT_RELU if relu else ACT_NONE, flags=flags, ld=w.stride(0), p0=w, p1=bias, p2=out,
                         ld2=0 if out is None else self._rows2d(out).stride(0),
                         reserved=groups if groups > 1 else 0, div=(src_stride << 16) | dst_stride)

    def layernorm(self, src, src_col, dst, dst_col, weight, bias, eps=1e-5):
        return self._add(op=OP_LAYERNORM, src=src, dst=dst, src_col=src_col, dst_col=dst_col, n=weight.shape[0], f0=eps,
                         p0=weight, p1=bias)

    def add(self, src, src_col, dst, dst_col, n):
        return self._add(op=OP_ADD, src=src, dst=dst, src_col=src_col, dst_col=dst_col, n=n)

    def copy(self, src, src_col, dst, dst_col, n):
        return self._add(op=OP_COPY, src=src, dst=dst, src_col=src_col, dst_col=dst_col, n=n)

    def clamp(self, dst, dst_col, n, lo, hi):
        return self._add(op=OP_CLAMP, dst=dst, dst_col=dst_col, n=n, f0=lo, f1=hi)

    def dropout(self, dst, dst_col, n, p: float, seed, site: int, step: int):
        """dst[:, dst_col:+n] in place with tbx_keyed_dropout's mask of (seed, site, step, global row, column of n)."""
        th = p * 4294967296.0
        th = 1 if 0 < th < 1 else int(th)
        return self._add(op=OP_DROPOUT, dst=dst, dst_col=dst_col, n=n, k=int(step), div=int(site), f0=1.0 / (1.0 - p),
                         reserved=th - (1 << 32) if th >= (1 << 31) else th, p0=seed)

    def rowmask(self, dst, dst_col, n, mask=None, fill=0.0, row_div=0, valid_mask=False):
        """Fill rows whose mask byte is set (valid_mask: whose byte is clear, i.e. `mask` is a validity array)."""
        flags, div = (F_ROW_DIV, row_div) if row_div else (0, 0)
        flags |= F_MASK_INV if valid_mask else 0
        return self._add(op=OP_ROWMASK, dst=dst, dst_col=dst_col, n=n, f0=fill, flags=flags, div=div, p0=mask)

    def groupmax(self, src, src_col, dst, dst_col, n, mask=None):
        """mask u8 [rows]: masked rows stay out of the maximum and are zeroed in the src and dst columns (see include/tbx_hip.h)."""
        return self._add(op=OP_GROUPMAX, src=src, dst=dst, src_col=src_col, dst_col=dst_col, n=n, p1=mask)

    def poolmax(self, src, src_col, n, out, out_col=0, mask=None, keep=None):
        """out[group] = max over the group's unmasked rows. keep=(buf, col): the pooled rows also stay in LDS (row j of `buf` != src =
        group j of the tile) and the stages after this one run on them - the tile's global rows are then its group indices."""
        if keep is None:
            return self._add(op=OP_POOLMAX, src=src, src_col=src_col, n=n, dst_col=out_col, ld=self._rows2d(out).stride(0), p0=out,
                             p1=mask)
        assert keep[0] != src and not self.live_rows
        return self._add(op=OP_POOLMAX, src=src, src_col=src_col, n=n, dst_col=out_col, ld=self._rows2d(out).stride(0), p0=out,
                         p1=mask, dst=keep[0], k=keep[1], flags=F_POOL_KEEP)

    def store(self, src, src_col, n, out, out_col=0):
        """out[g, out_col:+n] = src[:, src_col:+n]; a bfloat16 `out` receives the values rounded to nearest even."""
        return self._add(op=OP_STORE, src=src, src_col=src_col, n=n, dst_col=out_col, ld=self._rows2d(out).stride(0), p0=out,
                         flags=F_OUT_BF16 if out.dtype == torch.bfloat16 else 0)

    def store_masked_sum(self, src, src_col, n, group_stride, masks, out, out_col=0):
        """out[g, out_col:+n] = sum over the G groups i with masks[i, g] == 0 of src[:, src_col + i*group_stride : +n] (masks u8 [G, rows])."""
        assert masks.dtype == torch.uint8 and masks.dim() == 2 and masks.is_contiguous() and out.dtype == torch.float32
        return self._add(op=OP_STORE, src=src, src_col=src_col, n=n, dst_col=out_col, ld=self._rows2d(out).stride(0), p0=out, p1=masks,
                         reserved=masks.shape[0], div=group_stride, k=masks.shape[1], flags=F_MASKED_SUM)

    def run(self, n_rows: int, group_rows: int = 0):
        if self._arr is None:
            assert len(self.stages) <= MAX_STAGES, f"{len(self.stages)} stages > {MAX_STAGES}"
            self._arr = (Stage * len(self.stages))(*self.stages)
        if self.live_rows:
            assert group_rows == 0, "live-row chains are flat"
            rc = load().tbx_rowchain_live(self._arr, len(self.stages), n_rows, self.live_rows, self.ldw, self.ldw1, self.ld_aux,
                                          stream_ptr())
        else:
            rc = load().tbx_rowchain_ex(self._arr, len(self.stages), n_rows, group_rows, self.tile_rows, self.ldw, self.ldw1,
                                        self.ld_aux, stream_ptr())
        _check(rc, "tbx_rowchain")
