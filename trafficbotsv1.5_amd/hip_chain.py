"""The `Chain` builder of tbx_rowchain programs (row-tile interpreter: LINEAR / LayerNorm / ReLU / masks / group max / pooling stages over
LDS-resident activations; include/tbx_hip.h). Re-exported by hip.py."""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from .abi import *  # noqa: F401,F403  (constants, structures, load, declared_symbols: the C-ABI mirror)
from .abi import load  # noqa: F401
from .hip_base import _check, _cptr, _ptr, packed_weight, stream_ptr


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
