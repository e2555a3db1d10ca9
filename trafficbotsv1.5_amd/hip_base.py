"""ctypes glue shared by the hip_* wrapper modules: return-code check, device-pointer helpers, the current stream, the attention
segment descriptor and the weight-image caches (tbx_pack_weight* images per parameter version, or per training step inside PACK_SCOPE).
There is no CPU path: every helper raises on tensors that are not on a HIP device."""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import load  # noqa: F401


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


DEFERRED = None  # hip.defer(): the list that collects launch descriptors instead of launching them


def stream_ptr() -> int:
    if DEFERRED is not None:  # a launch that has no deferred form inside a one-queue step would run out of order
        raise RuntimeError("a kernel launch inside hip.defer() that is neither tbx_front nor tbx_knarpe_dec_layer")
    return torch.cuda.current_stream().cuda_stream


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


# Set to a dict for the duration of a training step (train_graph.training_step): chain kernels of the step's no-grad stepping
# pass then pack each weight ONCE PER STEP into this scope instead of the per-parameter cache. A captured training step
# (GraphedTrainStep) replays after the optimizer has moved the weights: a cached image from before the capture would be read by
# the replay without ever being re-packed (its tbx_pack_weight launch is not in the graph) - with the scope the packing is.
PACK_SCOPE: Optional[dict] = None


def _pack_key(w, bias, wt, groups, split, gemv, mfma32):
    """-> (cache dict, key, stamp) of a packed_weight request (see there)."""
    base = w._base if w._base is not None else w
    bkey = None if bias is None else (bias.data_ptr(), bias.shape[0])
    key = (w.storage_offset(), tuple(w.shape), w.stride(0), wt, groups, bkey, split, gemv, mfma32)
    if PACK_SCOPE is not None:  # a training step: images live (and are re-packed) per step, see PACK_SCOPE
        cache, key = PACK_SCOPE, (id(base),) + key
        PACK_SCOPE.setdefault("_keep", {})[id(base)] = base  # ids stay unique while the scope lives
    else:
        cache = base.__dict__.setdefault("_tbx_packed", {})
    return cache, key, (w._version, w.data_ptr(), None if bias is None else bias._version)


def packed_weights_mfma32_multi(reqs) -> None:
    """reqs = [(w, bias | None, wt, groups)]: the tbx_pack_weight_mfma32 images of all of them in ONE launch
    (tbx_pack_weight_mfma32_multi), left in the cache packed_weight(.., mfma32=True) looks them up in. Requests whose image is current
    are skipped."""
    from .abi import PackJob

    lib, jobs, keep = load(), [], []
    for w, bias, wt, groups in reqs:
        assert w.is_cuda and w.dim() == 2 and w.stride(1) == 1 and w.dtype == torch.float32
        cache, key, stamp = _pack_key(w, bias, wt, groups, False, False, True)
        hit = cache.get(key)
        if hit is not None and hit[0] == stamp:
            continue
        n, k = (w.shape[1], w.shape[0] // groups) if wt else (w.shape[0] // groups, w.shape[1])
        size = lib.tbx_pack_weight_mfma32_size(n, k, groups)
        if size <= 0:
            _check(int(size), "tbx_pack_weight_mfma32_size")
        if bias is not None:
            assert bias.is_cuda and bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == groups * n
        out = torch.empty(size, dtype=torch.float32, device=w.device)
        j = PackJob()
        j.w, j.bias, j.out, j.n, j.k, j.ld, j.groups, j.wt = _ptr(w), _ptr(bias), _ptr(out), n, k, w.stride(0), groups, int(wt)
        jobs.append(j)
        keep.append((cache, key, stamp, out))
    if not jobs:
        return
    arr = (PackJob * len(jobs))(*jobs)
    _check(lib.tbx_pack_weight_mfma32_multi(arr, len(jobs), stream_ptr()), "tbx_pack_weight_mfma32_multi")
    for cache, key, stamp, out in keep:
        cache[key] = (stamp, out)


def pack_group(tensors, reqs) -> None:
    """Inside a PACK_SCOPE: the first packed_weight(.., mfma32=True) request for any of `tensors` packs ALL of `reqs` in one launch (the
    folded weights of an attention module: its three LINEARs' images and the W^T images of their input gradients)."""
    if PACK_SCOPE is None:
        return
    g = PACK_SCOPE.setdefault("_groups", {})
    keep = PACK_SCOPE.setdefault("_keep", {})
    reqs = list(reqs)
    for t in tensors:
        g[id(t)] = reqs
        keep[("group", id(t))] = t


# The image requests of a training step whose sources are nn.Parameters (ready when the step starts) are kept, per owner (the model:
# the list dies with it) and key, as recorded during the previous step: open_pack_scope packs all of them in ONE launch (a 16-scene step
# asked for ~80 of them one by one, forward and backward).
def open_pack_scope(owner=None, plan_key=None) -> dict:
    """PACK_SCOPE = a new scope; with a list recorded for (owner, plan_key), its images first (one launch). -> the scope."""
    global PACK_SCOPE
    plans = None if owner is None else owner.__dict__.setdefault("_tbx_pack_plans", {})
    PACK_SCOPE = {"_plans": plans, "_plan_key": plan_key, "_record": {}}
    plan = plans.get(plan_key) if plans is not None else None
    if plan:
        plan = [r for r in plan if r[0].is_cuda and (r[1] is None or r[1].is_cuda)]  # (a model moved off the device since: nothing to pre-pack)
        packed_weights_mfma32_multi(plan)
        for r in plan:
            PACK_SCOPE["_record"][_pack_key(r[0], r[1], r[2], r[3], False, False, True)[1]] = r
    return PACK_SCOPE


def close_pack_scope(scope: Optional[dict] = None) -> None:
    """PACK_SCOPE = None; the scope's Parameter-sourced requests become the list the next scope of its owner and key starts from."""
    global PACK_SCOPE
    scope = PACK_SCOPE if scope is None else scope
    if scope is not None and scope.get("_plans") is not None:
        scope["_plans"][scope["_plan_key"]] = list(scope["_record"].values())
    PACK_SCOPE = None


def packed_weight(w: torch.Tensor, bias: Optional[torch.Tensor] = None, wt: bool = False, groups: int = 1,
                  split: bool = False, gemv: bool = False, mfma32: bool = False) -> torch.Tensor:
    """tbx_pack_weight image of a LINEAR weight (+ bias). Cached on the weight's base tensor object (the nn.Parameter)
    per view and version of both tensors: re-packed after an in-place update (optimizer step, load_state_dict), reused
    otherwise - chains are rebuilt every eager step - and dropped with the parameter.
    split=True: the tbx_pack_weight_split image (bf16 hi + lo halves) for stages flagged F_WSPLIT.
    gemv=True: the tbx_pack_weight_gemv image (column streams) for the F_WGEMV stages of live-row chains.
    mfma32=True: the tbx_pack_weight_mfma32 image (per-wave units of bf16 hi + lo fragments) for tbx_layer_tile."""
    assert w.is_cuda and w.dim() == 2 and w.stride(1) == 1 and w.dtype == torch.float32
    cache, key, stamp = _pack_key(w, bias, wt, groups, split, gemv, mfma32)
    if mfma32 and PACK_SCOPE is not None and "_record" in PACK_SCOPE:
        base = w._base if w._base is not None else w
        if isinstance(base, torch.nn.Parameter) and (bias is None or isinstance(bias if bias._base is None else bias._base, torch.nn.Parameter)):
            PACK_SCOPE["_record"].setdefault(key, (w, bias, wt, groups))  # (open_pack_scope: next step's plan)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    if mfma32 and PACK_SCOPE is not None:
        grp = PACK_SCOPE.get("_groups", {}).get(id(w))
        if grp is not None and any(r[0] is w and r[1] is bias and bool(r[2]) == bool(wt) and r[3] == groups for r in grp):
            packed_weights_mfma32_multi(grp)  # pack_group: this request's image and its siblings' in one launch
            return cache[key][1]
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
