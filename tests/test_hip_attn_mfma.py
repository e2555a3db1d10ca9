"""tbx_knarpe_attn_fwd_mfma (csrc/attn_mfma.hip: the wave-per-row KNARPE attention on the bf16 matrix cores) against
  (a) an fp64 torch evaluation of the factorised formula of include/tbx_hip.h (K6) on the same gathered rows, and
  (b) tbx_knarpe_attn_fwd (the fp32 VALU kernel, itself checked against the oracle / the reference's golden in test_hip_parity.py)
on random tables, ragged masks (rows without a valid target, segments shorter than a 16-target tile, 1 and 2 segments, tables
shared across batch entries), fp32 and bfloat16 tables.
Tolerance: 3e-2 absolute on outputs of O(1) (bf16 operands - 8-bit mantissas on q, k, v, e and the softmax weights - with fp32
accumulation and an fp32 softmax: the bf16-arithmetic schedule); flags and all-invalid rows exact."""
import numpy as np
import pytest
import torch

from oracle import hptr_ops as H

pytestmark = pytest.mark.gpu


def _case(g, n, S, segs_spec, bf16):
    """segs_spec: [(n_tgt, k, batch_div)] -> qbuf, segs (host tensors)"""
    d = 128
    qbuf = torch.randn(n * S, 896, generator=g) * 0.5  # [q | k | v | qt]: q at 0, qt at 384
    qbuf[:, 384:] *= 0.3
    segs = []
    for n_tgt, k, div in segs_spec:
        kv = torch.randn((n // div) * n_tgt, 2 * d, generator=g)
        idx = torch.stack([torch.randperm(n_tgt, generator=g)[:k] for _ in range(n * S)]).view(n, S, k).to(torch.int32)
        inv = torch.rand(n, S, k, generator=g) < 0.3
        rel = torch.cat([(torch.rand(n, S, k, 2, generator=g) - 0.5) * 300, (torch.rand(n, S, k, 1, generator=g) - 0.5) * 7], -1)
        segs.append(dict(kv=kv.to(torch.bfloat16) if bf16 else kv, n_tgt=n_tgt, k=k, div=div, idx=idx, inv=inv, rel=rel))
    # rows without a valid target at all, and a row with exactly one
    for s in segs:
        s["inv"][0, 1] = True
        s["inv"][-1, 0] = True
    segs[0]["inv"][-1, 0, 0] = False
    return qbuf, segs


def _reference(qbuf, segs, n, S):
    """fp64: score[h,t] = q_h . k_h + qt_h . e_t, softmax over valid targets of ALL segments, out = [sum a v | sum a e per head]."""
    fxy, fyw = H.make_freqs_xy(32, 1e3), H.make_freqs_rad(64)
    q = qbuf[:, :128].double().view(n * S, 4, 32)
    qt = qbuf[:, 384:896].double().view(n * S, 4, 128)
    sc, vs, es, ms = [], [], [], []
    for s in segs:
        kv = s["kv"].double().view(n // s["div"], s["n_tgt"], 256)
        b = torch.arange(n).repeat_interleave(S) // s["div"]
        rows = kv[b[:, None], s["idx"].view(n * S, -1).long()]  # [n*S, k, 256]
        k_, v_ = rows[..., :128].view(n * S, -1, 4, 32), rows[..., 128:].view(n * S, -1, 4, 32)
        e = H.pe_xy_yaw(s["rel"][..., :2].double(), s["rel"][..., 2].double(), fxy.double(), fyw.double()).view(n * S, -1, 128)
        sc.append(torch.einsum("rhc,rthc->rht", q, k_) + torch.einsum("rhc,rtc->rht", qt, e))
        vs.append(v_), es.append(e), ms.append(s["inv"].view(n * S, -1))
    sc, m = torch.cat(sc, -1) / np.sqrt(32.0), torch.cat(ms, -1)
    sc = sc.masked_fill(m[:, None, :], float("-inf"))
    none = m.all(-1)
    a = torch.softmax(sc.masked_fill(none[:, None, None], 0.0), -1).masked_fill(none[:, None, None], 0.0)
    ov = torch.einsum("rht,rthc->rhc", a, torch.cat(vs, 1)).reshape(n * S, 128)
    oe = torch.einsum("rht,rtc->rhc", a, torch.cat(es, 1)).reshape(n * S, 512)
    return torch.cat([ov, oe], -1).float(), none


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("n,S,spec", [(4, 300, [(90, 25, 1)]),                     # self attention of a step: one segment, 25 targets
                                      (4, 300, [(1024, 64, 4), (128, 25, 1)]),      # cross: map tokens shared by 4 entries + lights
                                      (2, 600, [(200, 7, 1), (64, 37, 2)]),         # segments shorter / longer than a tile, ragged
                                      (1, 1100, [(256, 128, 1)])])                  # the 128-target limit
def test_mfma_attention_vs_fp64_and_valu_kernel(tb, bf16, n, S, spec):
    from importlib import import_module

    hip = import_module("trafficbots_amd.hip")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1000 * S + len(spec) + int(bf16))
    qbuf, segs = _case(g, n, S, spec, bf16)
    ref, none = _reference(qbuf, segs, n, S)
    fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
    qd = qbuf.to(dev)
    hs = [hip.Seg(s["kv"].to(dev), 0, 128, s["n_tgt"], s["idx"].to(dev), s["inv"].to(torch.uint8).to(dev), None, s["div"],
                  rel=s["rel"].to(dev).contiguous()) for s in segs]
    bias0 = torch.zeros(128, device=dev)
    out_v = torch.full((n * S, 640), float("nan"), device=dev)
    flag_v = torch.empty(n * S, dtype=torch.uint8, device=dev)
    hip.knarpe_attn(qd, 0, 384, bias0, n, S, hs, out_v, flag_v, fxy, fyw)
    torch.testing.assert_close(out_v.cpu(), ref, rtol=1e-4, atol=1e-4)  # (the harness itself: the VALU kernel on these inputs)
    atol = 3e-2
    out = torch.full((n * S, 640), float("nan"), device=dev)
    flag = torch.full((n * S,), 7, dtype=torch.uint8, device=dev)
    hip.knarpe_attn_mfma(qd, 0, 384, n, S, hs, out, flag, fxy, fyw)
    torch.cuda.synchronize()
    assert torch.equal(flag.cpu().bool(), none)
    o = out.cpu()
    assert bool(torch.isfinite(o).all())
    assert float(o[none].abs().max()) == 0.0  # rows without a valid target: exact zeros
    err = (o - ref).abs().max().item()
    assert err < atol, err
    torch.testing.assert_close(o, out_v.cpu(), rtol=0, atol=atol)
    # the error is that of bf16 operands, not a layout slip: an order of magnitude below the tolerance in the mean
    assert (o - ref).abs().mean().item() < 3e-3
