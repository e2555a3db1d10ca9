"""ctypes wrappers of the TRAINING entry points of libtbx_hip.so (include/tbx_hip.h): keyed dropout and the one-pass glue ops, the tall
LINEAR / weight-gradient kernels of the time-batched pass, LayerNorm / PointNet-tail / masked-max-pool forward + backward, the attention
backward (atomics and inverse-list forms) and the per-step state machine (tbx_train_chain_*). Re-exported by hip.py."""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import load  # noqa: F401
from .hip_base import Seg, _check, _cptr, _drop_args, _ptr, packed_weight, stream_ptr


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
    multiples of 64 up to 1024, 16-byte aligned, leading dimension a multiple of 4."""
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == k and k % 64 == 0 and n % 64 == 0 and k <= 1024 and n <= 1024
            and x.stride(-1) == 1 and x.data_ptr() % 16 == 0)


def _pad128(w: torch.Tensor, b: Optional[torch.Tensor]):
    """(w, b) zero-padded to multiples of 128 in both dimensions (tbx_tall_linear's image for a 64-wide layer); cached like packed_weight:
    per parameter version, or per training step inside PACK_SCOPE."""
    from . import hip_base

    n, k = w.shape
    npad, kpad = -(-n // 128) * 128, -(-k // 128) * 128
    if (npad, kpad) == (n, k):
        return w, b
    key = ("pad128", id(w), None if b is None else id(b))
    stamp = (w._version, w.data_ptr(), None if b is None else b._version)
    cache = hip_base.PACK_SCOPE if hip_base.PACK_SCOPE is not None else w.__dict__.setdefault("_tbx_pad128", {})
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    with torch.no_grad():
        wp = torch.zeros(npad, kpad, dtype=torch.float32, device=w.device)
        wp[:n, :k].copy_(w)
        bp = None
        if b is not None:
            bp = torch.zeros(npad, dtype=torch.float32, device=w.device)
            bp[:n].copy_(b)
    cache[key] = (stamp, wp, bp)
    if hip_base.PACK_SCOPE is not None:
        hip_base.PACK_SCOPE.setdefault("_keep", {})[id(w)] = w
    return wp, bp


def tall_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, wt: bool = False, relu: bool = False,
                bf16: bool = False, out: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None, drop=None) -> torch.Tensor:
    """y = x W^T (+ b) over many rows on the split-bf16 matrix path (tbx_tall_linear; bf16: ONE bf16 product per term,
    tbx_tall_linear_bf16). wt: w is stored [k x n] (the input gradient dx = dy W of a Linear with weight W [n_out, n_in]: x = dy,
    w = W, wt = True). drop (with relu): tbx_keyed_dropout's arguments - dropout(relu(.)) in the same launch."""
    n, k = (w.shape[1], w.shape[0]) if wt else (w.shape[0], w.shape[1])
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    wp, bp = _pad128(w, b)  # (a 64-wide layer: the image of the zero-padded weight; k / n below stay the valid widths)
    img = packed_weight(wp, bp, wt=wt, mfma32=True)
    y = torch.empty(x2.shape[0], n, dtype=torch.float32, device=x.device) if out is None else out
    # (out: rows may be a column block of a wider table - leading dimension a multiple of 4 floats, 16-byte aligned)
    assert y.shape == (x2.shape[0], n) and y.stride(1) == 1 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0 and y.dtype == torch.float32
    ldy = y.stride(0)
    if out16 is not None:  # the same rows as bfloat16 as well (tbx_tall_linear_dual)
        assert out16.shape == y.shape and out16.dtype == torch.bfloat16 and out16.is_contiguous()
        fn = load().tbx_tall_linear_dual_bf16 if bf16 else load().tbx_tall_linear_dual
        _check(fn(_ptr(x2, torch.float32), x2.shape[0], k, x2.stride(0), _ptr(img, torch.float32), n, int(b is not None), int(relu),
                  _ptr(y), ldy, _ptr(out16, torch.bfloat16), n, stream_ptr()), "tbx_tall_linear_dual")
        return y if out is not None else y.view(*x.shape[:-1], n)
    if drop is not None:  # dropout(relu(.)) in the launch (tbx_tall_linear_relu_drop): drop = (p, seed, site, rows_per_scene, time_batch, time0)
        assert relu
        p, seed, site, rps, tb, t0 = _drop6(drop)
        fn = load().tbx_tall_linear_relu_drop_bf16 if bf16 else load().tbx_tall_linear_relu_drop
        _check(fn(_ptr(x2, torch.float32), x2.shape[0], k, x2.stride(0), _ptr(img, torch.float32), n, int(b is not None), _ptr(y), ldy,
                  p, seed, site, rps, tb, t0, stream_ptr()), "tbx_tall_linear_relu_drop")
        return y if out is not None else y.view(*x.shape[:-1], n)
    fn = load().tbx_tall_linear_bf16 if bf16 else load().tbx_tall_linear
    _check(fn(_ptr(x2, torch.float32), x2.shape[0], k, x2.stride(0), _ptr(img, torch.float32), n, int(b is not None), int(relu),
              _ptr(y), ldy, stream_ptr()), "tbx_tall_linear")
    return y if out is not None else y.view(*x.shape[:-1], n)


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


def attn_fold_fwd(w, b, wr, br, wo, bo):
    """tbx_attn_fold_fwd: (w_in [640,128], b_in [640], w_kv [256,128], b_kv [256], bias_k [128], w_out [128,640], b_out [128]) of one
    AttentionRPE module from in_proj (w [384,128], b [384]), linear_rpe (wr [256,128], br [256]) and out_proj (wo [128,128], bo [128])."""
    for t, shp in ((w, (384, 128)), (b, (384,)), (wr, (256, 128)), (br, (256,)), (wo, (128, 128)), (bo, (128,))):
        assert t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == shp and t.is_contiguous(), shp
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=w.device)
    out = (e(640, 128), e(640), e(256, 128), e(256), e(128), e(128, 640), e(128))
    _check(load().tbx_attn_fold_fwd(_ptr(w), _ptr(b), _ptr(wr), _ptr(br), _ptr(wo), _ptr(bo), *[_ptr(t) for t in out], stream_ptr()), "tbx_attn_fold_fwd")
    return out


def attn_fold_bwd(w, b, wr, br, wo, grads):
    """tbx_attn_fold_bwd: grads = the seven output gradients (None = zero) -> (d_w, d_b, d_wr, d_br, d_wo, d_bo)."""
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=w.device)
    out = (e(384, 128), e(384), e(256, 128), e(256), e(128, 128), e(128))
    g = [None if t is None else t.contiguous() for t in grads]
    _check(load().tbx_attn_fold_bwd(_ptr(w), _ptr(b), _ptr(wr), _ptr(br), _ptr(wo), *[_cptr(t, torch.float32) for t in g], *[_ptr(t) for t in out],
                                    stream_ptr()), "tbx_attn_fold_bwd")
    return out


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


def pair_bias_relu(h, pa, pm, relu: bool):
    """h [n, A, M, d] += pa [n, A, d] + pm [n, M, d] (broadcast), relu'd if `relu`, in place (tbx_pair_bias_relu)."""
    n, A, M, d = h.shape
    assert glue_ok(h) and h.is_contiguous() and pa.shape == (n, A, d) and pm.shape == (n, M, d)
    pa, pm = pa.contiguous(), pm.contiguous()
    _check(load().tbx_pair_bias_relu(_ptr(h, torch.float32), _ptr(pa, torch.float32), _ptr(pm, torch.float32), n, A, M, d, int(bool(relu)), stream_ptr()),
           "tbx_pair_bias_relu")
    return h


def pointnet_tail_ok(z: torch.Tensor) -> bool:
    """z [G, W, 64] fp32 on the device, W <= 32: the shapes tbx_pointnet_tail_* / tbx_masked_maxpool_* take."""
    return z.is_cuda and z.dtype == torch.float32 and z.dim() == 3 and z.shape[2] == 64 and 0 < z.shape[1] <= 32 and z.shape[0] > 0


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


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, add: Optional[torch.Tensor] = None):
    """(dx, dgamma, dbeta) of y = LayerNorm_128(x) * gamma + beta given dy, with the forward's per-row mean / rstd (tbx_layernorm_bwd).
    add [rows, 128]: dx = add + that gradient in the same pass (tbx_layernorm_bwd_add: the residual branch's gradient of x)."""
    assert layernorm_bwd_ok(x) and x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and dy.dtype == torch.float32
    assert add is None or (add.is_contiguous() and add.shape == x.shape and add.dtype == torch.float32)
    rows = x.numel() // 128
    assert mean.numel() == rows and rstd.numel() == rows and mean.is_contiguous() and rstd.is_contiguous()
    lib = load()
    n = lib.tbx_layernorm_bwd_partials(rows)
    scratch = torch.empty(n, 256, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dg = torch.empty(128, dtype=torch.float32, device=x.device)
    db = torch.empty(128, dtype=torch.float32, device=x.device)
    _check(lib.tbx_layernorm_bwd_add(_ptr(x), _ptr(dy), _ptr(gamma.contiguous(), torch.float32), _ptr(mean, torch.float32), _ptr(rstd, torch.float32),
                                     rows, 128, _ptr(add), _ptr(dx), _ptr(dg), _ptr(db), _ptr(scratch), stream_ptr()), "tbx_layernorm_bwd_add")
    return dx, dg, db


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
