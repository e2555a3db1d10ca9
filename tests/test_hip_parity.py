"""GPU parity tests proper: the HIP path (called through the C ABI via hip.py / the reference-named modules) against
the oracle on identical seeded inputs. Bit-exact for K-nearest index sets and masks; fp32 tolerance stated per check.
Run on the MI355X box with `pytest -m gpu`."""
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import hptr_ops as H
from oracle import trafficbots_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-4, atol=2e-5)  # fp32 kernels vs fp32 CPU oracle (different summation order, ocml vs sleef sin/cos)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip(tb):
    h = import_module("trafficbots_amd.hip")
    h.load()
    return h


def _poses(g, n, S, span=200.0):
    return torch.cat([(torch.rand(n, S, 2, generator=g) - 0.5) * span, (torch.rand(n, S, 1, generator=g) - 0.5) * 6.28], -1)


@pytest.mark.parametrize("S,T,K,limit", [(12, 40, 6, 80.0), (64, 1024, 64, 500.0), (128, 128, 24, 250.0), (5, 2000, 33, 1e9),
                                         (2100, 40, 6, 80.0)])  # last: >= 4096 rows -> the one-wave-per-row kernel variant
def test_knn_embed(hip, dev, S, T, K, limit):
    g = torch.Generator().manual_seed(S * 1000 + T)
    n = 2
    pose, pose2 = _poses(g, n, S), _poses(g, n, T)
    inv, inv2 = torch.rand(n, S, generator=g) < 0.2, torch.rand(n, T, generator=g) < 0.3
    inv2[1, K // 2:] = True  # fewer valid targets than K in scene 1
    rp, rd = H.rel_pose(pose, inv, pose2, inv2)
    idx_o, inv_o, rpe_o = H.knn_select(inv2, rp, rd, K, limit)
    fxy, fyw = H.make_freqs_xy(32, 1e3), H.make_freqs_rad(64)
    idx, kinv, rel, emb = hip.knn_embed(pose.to(dev), inv.to(torch.uint8).to(dev), pose2.to(dev), inv2.to(torch.uint8).to(dev), K,
                                        limit, fxy.to(dev), fyw.to(dev), 128, want_rel_pose=True)
    assert torch.equal(H.sorted_valid_sets(idx.cpu(), kinv.cpu()), H.sorted_valid_sets(idx_o, inv_o))  # bit-exact sets
    assert int(idx.min()) >= 0 and int(idx.max()) < T
    # relative poses / embeddings of the valid neighbours, matched by index
    rel_full = torch.gather(rp, 2, idx.cpu().long()[..., None].expand(-1, -1, -1, 3))
    ok = ~kinv.cpu().bool()
    torch.testing.assert_close(rel.cpu()[ok], rel_full[ok], rtol=1e-5, atol=1e-4)
    e_ref = H.pe_xy_yaw(rel.cpu()[..., :2], rel.cpu()[..., 2], fxy, fyw)
    torch.testing.assert_close(emb.cpu()[ok], e_ref[ok], rtol=1e-4, atol=2e-5)


def test_knn_lattice_exact(hip, dev):
    """Integer-lattice poses: every distance is exact in fp32, so the sets must agree even at near-ties."""
    g = torch.Generator().manual_seed(5)
    lat = torch.cat([torch.randint(-40, 40, (1, 200, 2), generator=g).float() * 0.25,
                     torch.randint(0, 4, (1, 200, 1), generator=g).float() * (np.pi / 2)], -1)
    inv = torch.zeros(1, 200, dtype=torch.bool)
    rp, rd = H.rel_pose(lat, inv)
    K = 9
    idx, kinv, _, _ = hip.knn_embed(lat.to(dev), inv.to(torch.uint8).to(dev), lat.to(dev), inv.to(torch.uint8).to(dev), K, 1e9,
                                    want_emb=False)
    # compare the multiset of selected DISTANCES (ties at the K-th place may pick different equal-distance indices)
    d_hip = torch.gather(rd, 2, idx.cpu().long()).sort(-1)[0]
    d_ref = torch.topk(rd, K, dim=-1, largest=False)[0].sort(-1)[0]
    torch.testing.assert_close(d_hip, d_ref, rtol=0, atol=2e-5)


def _filled(tb, cls, seed, dev, **kw):
    m = cls(**kw)
    tb.utils.det_fill(m, seed)
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(dev).eval(), P


def test_mlp_input_pointnet(tb, hip, dev):
    M = import_module("trafficbots_amd.models.modules")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 7, 20, generator=g)
    inv = torch.rand(2, 5, 7, generator=g) < 0.4
    inv[0, 0] = True
    m, P = _filled(tb, M.mlp.MLP, 11, dev, fc_dims=[20, 64, 64, 64], end_layer_activation=False)
    P = {"m." + k: v for k, v in P.items()}
    torch.testing.assert_close(m(x.to(dev)).cpu(), H.mlp(P, "m", x, end_act=False), **TOL)
    torch.testing.assert_close(m(x.to(dev), inv.to(dev), float("-inf")).cpu(),
                               H.mlp(P, "m", x, end_act=False, mask_invalid=inv, fill=float("-inf")), **TOL)
    m, P = _filled(tb, M.mlp.MLP, 12, dev, fc_dims=[48, 32, 32, 1], end_layer_activation=False, use_layernorm=True)
    P = {"m." + k: v for k, v in P.items()}
    x2 = torch.randn(37, 48, generator=g)
    torch.testing.assert_close(m(x2.to(dev)).cpu(), H.mlp(P, "m", x2, end_act=False), **TOL)
    m, P = _filled(tb, M.mlp.MLP, 13, dev, fc_dims=[31, 121, 121, 121], end_layer_activation=False)  # unaligned K / N
    P = {"m." + k: v for k, v in P.items()}
    x3 = torch.randn(50, 31, generator=g)
    torch.testing.assert_close(m(x3.to(dev)).cpu(), H.mlp(P, "m", x3, end_act=False), **TOL)
    for mode, pe_dim in (("cat", 64), ("add", 128)):
        m, P = _filled(tb, M.input_encoder.InputEncoder, 13, dev, hidden_dim=128, attr_dim=20, pe_dim=pe_dim, n_layer=3,
                       mlp_dropout_p=0, mlp_use_layernorm=False, mode=mode)
        P = {"ie." + k: v for k, v in P.items()}
        pe = torch.randn(2, 5, 7, pe_dim, generator=g)
        torch.testing.assert_close(m(x.to(dev), pe.to(dev)).cpu(), H.input_encoder(P, "ie", mode, x, pe), **TOL)
    for n_node in (7, 11, 20):
        m, P = _filled(tb, M.polyline_encoder.PolylineEncoder, 14, dev, hidden_dim=128, tf_cfg={}, n_layer=3,
                       mlp_use_layernorm=False, mlp_dropout_p=0.1, use_pointnet=True, pooling_mode="max_valid")
        P = {"pn." + k: v for k, v in P.items()}
        xp = torch.randn(2, 5, n_node, 128, generator=g)
        ip = torch.rand(2, 5, n_node, generator=g) < 0.4
        ip[0, 0] = True
        y16 = m(xp.to(dev), ip.to(dev))
        torch.testing.assert_close(y16.cpu(), H.pointnet(P, "pn", xp, ip, 3), **TOL)
        # 32- / 48-row tiles hold floor(tile / n_node) whole polylines (10 polylines: the last tile is partial); same rows, same
        # arithmetic per row -> bit-identical
        for tile in (32, 48):  # 48 rows: 6 / 4 / 2 whole polylines per tile
            m.tile_rows = tile
            assert torch.equal(m(xp.to(dev), ip.to(dev)), y16), (n_node, tile)
        m.tile_rows = None


def _attn_inputs(g, n=2, S=9, K=11, d=128, Ks=5):
    src = torch.randn(n, S, d, generator=g)
    tgt = torch.randn(n, S, K, d, generator=g)
    rpe_e = torch.randn(n, S, K, d, generator=g)
    m = torch.rand(n, S, K, generator=g) < 0.3
    m[0, 2] = True
    m[1, 0] = True
    src_inv = torch.rand(n, S, generator=g) < 0.2
    idx_self = torch.randint(0, S, (n, S, Ks), generator=g)
    m_self = torch.rand(n, S, Ks, generator=g) < 0.3
    m_self[0, 1] = True
    rpe_self = torch.randn(n, S, Ks, d, generator=g)
    return src, tgt, rpe_e, m, src_inv, idx_self, m_self, rpe_self


@pytest.mark.parametrize("S,K", [(9, 11), (70, 89), (33, 128), (2100, 8)])  # last: one-wave-per-row variant
def test_attention_rpe(tb, hip, dev, S, K):
    M = import_module("trafficbots_amd.models.modules")
    g = torch.Generator().manual_seed(S)
    src, tgt, rpe_e, m, *_ = _attn_inputs(g, S=S, K=K)
    att, P = _filled(tb, M.attention_rpe.AttentionRPE, 15, dev, d_model=128, n_head=4, dropout_p=0.1, d_rpe=128)
    P = {"a." + k: v for k, v in P.items()}
    out, _ = att(src.to(dev), tgt.to(dev), tgt_padding_mask=m.to(dev), rpe=rpe_e.to(dev))
    ref = H.attention_rpe(P, "a", 4, src, tgt, m, rpe_e)
    torch.testing.assert_close(out.cpu(), ref, **TOL)
    assert float(out[0, 2].abs().max()) == 0.0 and float(out[1, 0].abs().max()) == 0.0  # all-invalid rows -> exact 0


def test_pose_embed(tb, hip, dev):
    g = torch.Generator().manual_seed(2)
    pose = torch.cat([(torch.rand(300, 2, generator=g) - 0.5) * 900, (torch.rand(300, 1, generator=g) - 0.5) * 12], -1)
    for dim in (128, 64):
        fxy, fyw = H.make_freqs_xy(dim // 4, 1e3), H.make_freqs_rad(dim // 2)
        e = hip.pose_embed(pose.to(dev), fxy.to(dev), fyw.to(dev), dim)
        torch.testing.assert_close(e.cpu(), H.pe_xy_yaw(pose[:, :2], pose[:, 2], fxy, fyw), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("mode", ["enc_self_attn", "dec_cross_attn"])
def test_transformer_block(tb, hip, dev, mode):
    M = import_module("trafficbots_amd.models.modules")
    g = torch.Generator().manual_seed(21)
    src, tgt, rpe_e, m, src_inv, idx_self, m_self, rpe_self = _attn_inputs(g, S=37, K=13, Ks=7)
    tf_cfg = dict(d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1, bias=True, activation="relu", out_layernorm=False,
                  apply_q_rpe=False)
    blk, P = _filled(tb, M.transformer_rpe.TransformerBlockRPE, 16, dev, n_layer=3, mode=mode, d_rpe=128, **tf_cfg)
    P = {"t." + k: v for k, v in P.items()}
    to = lambda t: t.to(dev)
    eng = import_module("trafficbots_amd.engine")
    # both arithmetic classes of the small-launch schedule against the oracle: every LINEAR as exact-fp32 products (atol 5e-5), and
    # the default - the decoder layer as one launch on the split-bf16 matrix path, < 3e-5 of sum |x||w| per LINEAR output, which
    # three layers of LayerNorm / softmax carry to < 1.5e-4 on rows of O(1) entries
    for sched, atol in ((eng.DEFAULT.replace(dec_tail_mfma=False, tile_small=False), 5e-5), (eng.DEFAULT, 1.5e-4)):
        with eng.use(sched):
            if mode == "enc_self_attn":
                y, _ = blk(src=to(src), src_padding_mask=to(src_inv), tgt=to(idx_self), tgt_padding_mask=to(m_self), rpe=to(rpe_self))
                ref = H.transformer_block(P, "t", mode, 3, 4, src, src_inv, idx_self, m_self, rpe_self)
            else:
                y, _ = blk(src=to(src), src_padding_mask=to(src_inv), tgt=to(tgt), tgt_padding_mask=to(m), rpe=to(rpe_e),
                           decoder_tgt=to(idx_self), decoder_tgt_padding_mask=to(m_self), decoder_rpe=to(rpe_self))
                ref = H.transformer_block(P, "t", mode, 3, 4, src, src_inv, tgt, m, rpe_e, idx_self, m_self, rpe_self)
        torch.testing.assert_close(y.cpu(), ref, rtol=5e-4, atol=atol)


def _model(tb, dev, n_tgt_knn):
    cfg = tb.config.default_model_cfg(n_tgt_knn=n_tgt_knn)
    M = import_module("trafficbots_amd.models.traffic_bots")
    model = M.TrafficBots(**cfg)
    tb.utils.det_fill(model, 0)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return cfg, model.to(dev).eval(), P


@pytest.mark.parametrize("sizes,knn,n_steps", [((8, 64, 8), 4, 14), ((64, 1024, 128), 32, 3)])
def test_model_tokens_and_policy_steps(tb, hip, dev, sizes, knn, n_steps):
    """Rows 8-14: map tokens, tl pre-compute KNN sets, and the per-step policy (reference `TrafficBots.forward` API)
    driven by ground-truth states, vs the oracle."""
    cfg, model, P = _model(tb, dev, knn)
    batch = tb.synthetic.make_scene(1, *sizes, seed=0)
    b = O.scene_centric({**batch, **tb.synthetic.to_history_batch(batch)}, training=False)
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["gt/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    mp = model.mp_encoder(bd["sc/mp_valid"], bd["sc/mp_attr"], bd["sc/mp_pose"], bd["ref/mp_type"])
    assert torch.equal(mp["mp_token_invalid"].cpu(), mp_o["mp_token_invalid"])
    torch.testing.assert_close(mp["mp_token_feature"].cpu(), mp_o["mp_token_feature"], rtol=1e-3, atol=1e-4)
    tl = model.tl_encoder.pre_compute(tl_valid=bd["gt/tl_valid"], tl_attr=bd["sc/tl_attr"], tl_pose=bd["sc/tl_pose"], **mp)
    assert torch.equal(H.sorted_valid_sets(tl["knn_idx_tl2tl"].cpu(), tl["knn_invalid_tl2tl"].cpu()),
                       H.sorted_valid_sets(tl_o["knn_idx_tl2tl"], tl_o["knn_invalid_tl2tl"]))
    assert torch.equal(H.sorted_valid_sets(tl["knn_idx_tl2mp"].cpu(), tl["knn_invalid_tl2mp"].cpu()),
                       H.sorted_valid_sets(tl_o["knn_idx_tl2mp"], tl_o["knn_invalid_tl2mp"]))
    # per-step policy on ground-truth states (open loop), same injected latent
    g = torch.Generator().manual_seed(1)
    n, A = b["gt/ag_valid"].shape[:2]
    z = torch.randn(n, A, 16, generator=g)
    z_valid = b["gt/ag_valid"].any(-1)
    model.init()
    om.init()
    for t in range(n_steps):
        args = dict(ag_valid=b["gt/ag_valid"][:, :, t], ag_pose=b["gt/ag_pose"][:, :, t], ag_motion=b["gt/ag_motion"][:, :, t],
                    ag_attr=b["sc/ag_attr"], ag_type=b["ref/ag_type"], ag_latent=z, ag_latent_valid=z_valid,
                    ag_navi=b["gt/ag_navi"], ag_navi_valid=z_valid, tl_state=b["gt/tl_state"][:, :, t])
        with torch.no_grad():
            mean_o, _, logit_o = om.forward(tl_tokens=tl_o, mp_tokens=mp_o, **args)
        a_dist, tl_dist = model(ag_navi_updated=True, tl_tokens=tl, mp_tokens=mp, **{k: v.to(dev) for k, v in args.items()})
        torch.testing.assert_close(a_dist.mean.cpu(), mean_o, rtol=2e-3, atol=2e-4)
        torch.testing.assert_close(tl_dist.logits.cpu(), torch.log_softmax(logit_o, -1), rtol=2e-3, atol=2e-4)


def test_attention_relative_pose_mode_matches_materialised(tb, hip, dev):
    """The attention kernel fed with 12-B relative poses (embedding rebuilt in registers through the hardware
    sin/cos with an exact two-constant range reduction) must match the same call fed with the materialised embedding
    (libm sincosf), also for the ~600 rad arguments of far-away targets."""
    M = import_module("trafficbots_amd.models.modules")
    g = torch.Generator().manual_seed(9)
    n, S, T, K, d = 2, 40, 300, 33, 128
    att, P = _filled(tb, M.attention_rpe.AttentionRPE, 15, dev, d_model=d, n_head=4, dropout_p=0.1, d_rpe=d)
    src_pose, tgt_pose = _poses(g, n, S, 900.0), _poses(g, n, T, 900.0)
    inv_s, inv_t = torch.zeros(n, S, dtype=torch.uint8), (torch.rand(n, T, generator=g) < 0.2).to(torch.uint8)
    fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
    idx, kinv, rel, emb = hip.knn_embed(src_pose.to(dev), inv_s.to(dev), tgt_pose.to(dev), inv_t.to(dev), K, 1e9, fxy, fyw, 128,
                                        want_rel_pose=True, want_emb=True)
    qbuf = torch.randn(n * S, 640, generator=g).to(dev)
    kv = torch.randn(n * T, 256, generator=g).to(dev)
    outs = []
    for mode in ("emb", "rel"):
        out = torch.empty(n * S, 640, device=dev)
        flag = torch.empty(n * S, dtype=torch.uint8, device=dev)
        seg = hip.Seg(kv, 0, 128, T, idx, kinv, emb if mode == "emb" else None, rel=rel if mode == "rel" else None)
        hip.knarpe_attn(qbuf, 0, 128, att.linear_rpe.bias, n, S, [seg], out, flag, fxy, fyw)
        outs.append(out)
    torch.testing.assert_close(outs[1], outs[0], rtol=2e-4, atol=2e-5)
    # the E-sums are convex combinations of embedding channels: their agreement bounds the per-channel sin/cos error
    assert float((outs[1][:, 128:] - outs[0][:, 128:]).abs().max()) < 1e-5


@pytest.mark.parametrize("rows,k,n,groups,wt", [(64, 128, 128, 1, False), (40, 20, 5, 1, False), (16, 512, 128, 1, False),
                                                (70, 32, 128, 4, True), (33, 128, 32, 4, False), (5000, 128, 384, 1, False)])
def test_packed_weight_image_equals_row_major_linear(hip, dev, rows, k, n, groups, wt):
    """tbx_pack_weight + TBX_F_WPACK feed the MFMAs the same operands in the same order as the row-major reference path
    of LINEAR: bit-identical outputs (odd k / n, grouped, [k,n] layout, >= 4096 rows), and both match torch."""
    g = torch.Generator().manual_seed(rows + k + n)
    x = torch.randn(rows, groups * k if groups > 1 and not wt else max(k, groups * k), generator=g)
    w = torch.randn(groups * (k if wt else n), n if wt else k, generator=g) * 0.1
    bias = torch.randn(groups * n, generator=g)
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    outs = []
    for pack in (True, False):
        out = torch.zeros(rows, groups * n, device=dev)
        ch = hip.Chain(16, 1028)
        ch.pack_weights = pack
        kw = max(k, 16)
        ch.load(xd, hip.BUF0, 0, n=xd.shape[1], pad_to=((xd.shape[1] + 15) // 16) * 16)
        ch.linear(hip.BUF0, 0, hip.BUF1, 0, wd, bd, wt=wt, groups=groups, src_stride=k if groups > 1 else 0,
                  dst_stride=n if groups > 1 else 0)
        ch.store(hip.BUF1, 0, groups * n, out)
        ch.run(rows)
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    xs = x[:, :groups * k].reshape(rows, groups, k)
    ws = w.reshape(groups, k, n) if wt else w.reshape(groups, n, k).transpose(1, 2)
    ref = torch.einsum("rgk,gkn->rgn", xs.double(), ws.double()).reshape(rows, groups * n) + bias.double()
    torch.testing.assert_close(outs[0].double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,k,n,groups,wt,tile", [(64, 128, 128, 1, False, 16), (40, 20, 5, 1, False, 16), (16, 512, 128, 1, False, 16),
                                                     (70, 32, 128, 4, True, 16), (33, 128, 32, 4, False, 16), (5000, 128, 384, 1, False, 32), (100, 64, 128, 2, False, 32)])
def test_split_bf16_linear_close_to_exact_fp32_linear(hip, dev, rows, k, n, groups, wt, tile):
    """TBX_F_WSPLIT (tbx_pack_weight_split image, three bf16 MFMA products with fp32 accumulation): within 3e-5 of the
    magnitude sum_k |x_k w_k| + |b| of the exact result (bf16 hi + lo keeps 16 mantissa bits of each operand; the dropped
    lo*lo term is 2^-18 of a product), opt-in through Chain.split_bf16 / TBX_SPLIT_BF16=1."""
    g = torch.Generator().manual_seed(rows + k + n)
    x = torch.randn(rows, groups * k if groups > 1 and not wt else max(k, groups * k), generator=g)
    w = torch.randn(groups * (k if wt else n), n if wt else k, generator=g) * 0.1
    bias = torch.randn(groups * n, generator=g)
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    out = torch.zeros(rows, groups * n, device=dev)
    ch = hip.Chain(tile, 1028 if tile == 16 else 388)  # LDS bound: (2 ldw + 260) * tile_rows * 4 <= 160 KiB
    ch.split_bf16 = True
    ch.load(xd, hip.BUF0, 0, n=xd.shape[1], pad_to=((xd.shape[1] + 15) // 16) * 16)
    ch.linear(hip.BUF0, 0, hip.BUF1, 0, wd, bd, wt=wt, groups=groups, src_stride=k if groups > 1 else 0,
              dst_stride=n if groups > 1 else 0)
    ch.store(hip.BUF1, 0, groups * n, out)
    ch.run(rows)
    xs = x[:, :groups * k].reshape(rows, groups, k).double()
    ws = (w.reshape(groups, k, n) if wt else w.reshape(groups, n, k).transpose(1, 2)).double()
    ref = torch.einsum("rgk,gkn->rgn", xs, ws).reshape(rows, groups * n) + bias.double()
    mag = torch.einsum("rgk,gkn->rgn", xs.abs(), ws.abs()).reshape(rows, groups * n) + bias.double().abs()
    err = (out.cpu().double() - ref).abs()
    assert float((err / mag).max()) < 3e-5, float((err / mag).max())


@pytest.mark.parametrize("n", [32, 16])  # 2048 rows: a wave per row, plain gathers; 1024 rows: one wave per SIMD -> the LDS-ring kernel
@pytest.mark.parametrize("mode", ["rel", "emb", "rel_dropout"])
def test_attention_large_grid_kernel_matches_small_grid_kernel(tb, hip, dev, mode, n):
    """Grids of >= 1024 rows run one wavefront per row, smaller ones four wavefronts per row merged through LDS (checked
    against the oracle above): the same 2048 rows as one launch and as four 512-row launches must agree - two segments with
    shared tables (batch_div), masked pairs, a row without a valid target, in-register and materialised embeddings, and the
    dropout mask (keyed by the launch-local row index, hence compared on the first quarter only)."""
    g = torch.Generator().manual_seed(12)
    S, T1, T2, K1, K2, div = 64, 128, 96, 25, 9, 8
    rows = n * S
    P = import_module("trafficbots_amd.utils.pose_emb")
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    qbuf = torch.randn(rows, 640, generator=g).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    kv1 = torch.randn(n * T1, 256, generator=g).to(dev)
    kv2 = torch.randn((n // div) * T2, 512, generator=g).to(dev)  # shared by `div` batches, K | V inside a wider table
    idx1 = torch.randint(0, T1, (n, S, K1), generator=g).to(torch.int32).to(dev)
    idx2 = torch.randint(0, T2, (n, S, K2), generator=g).to(torch.int32).to(dev)
    inv1 = (torch.rand(n, S, K1, generator=g) < 0.3).to(torch.uint8)
    inv2 = (torch.rand(n, S, K2, generator=g) < 0.3).to(torch.uint8)
    inv1[3, 5], inv2[3, 5] = 1, 1  # a row without any valid target
    inv1, inv2 = inv1.to(dev), inv2.to(dev)
    rel1 = torch.cat([(torch.rand(n, S, K1, 2, generator=g) - 0.5) * 300, (torch.rand(n, S, K1, 1, generator=g) - 0.5) * 6.28], -1).to(dev)
    rel2 = torch.cat([(torch.rand(n, S, K2, 2, generator=g) - 0.5) * 300, (torch.rand(n, S, K2, 1, generator=g) - 0.5) * 6.28], -1).to(dev)
    fx, fy = pe.pe_xy.freqs, pe.pe_yaw.freqs
    emb = mode == "emb"
    if emb:
        e1 = pe(rel1[..., :2], rel1[..., 2:3]).contiguous()
        e2 = pe(rel2[..., :2], rel2[..., 2:3]).contiguous()
    drop = (0.2, torch.tensor([987654321012345], dtype=torch.int64, device=dev), 5) if mode == "rel_dropout" else None

    def run(b0, nb):
        sl = slice(b0, b0 + nb)
        seg1 = hip.Seg(kv1[b0 * T1:(b0 + nb) * T1], 0, 128, T1, idx1[sl].contiguous(), inv1[sl].contiguous(),
                       emb=e1[sl].contiguous() if emb else None, rel=None if emb else rel1[sl].contiguous())
        seg2 = hip.Seg(kv2[(b0 // div) * T2:((b0 + nb) // div) * T2], 128, 384, T2, idx2[sl].contiguous(), inv2[sl].contiguous(),
                       emb=e2[sl].contiguous() if emb else None, rel=None if emb else rel2[sl].contiguous(), batch_div=div)
        out = torch.empty(nb * S, 640, device=dev)
        flag = torch.empty(nb * S, dtype=torch.uint8, device=dev)
        hip.knarpe_attn(qbuf[b0 * S:(b0 + nb) * S], 0, 128, bias, nb, S, [seg1, seg2], out, flag, None if emb else fx, None if emb else fy,
                        drop=drop)
        return out, flag

    big, flag_big = run(0, n)                      # 2048 / 1024 rows: a wave per row
    parts = [run(b0, 8) for b0 in range(0, n, 8)]  # 512 rows each: four waves per row
    n_cmp = 1 if drop is not None else len(parts)  # the dropout counter uses the launch's own row numbering
    small = torch.cat([p[0] for p in parts[:n_cmp]])
    flag_small = torch.cat([p[1] for p in parts[:n_cmp]])
    assert torch.equal(flag_big[:small.shape[0]], flag_small)
    assert int(flag_big.sum()) == 1
    torch.testing.assert_close(big[:small.shape[0]], small, rtol=2e-4, atol=2e-5)


def test_action_head_fused_branches_bit_identical(tb, hip, dev):
    """ActionHead.emit with the three per-type branches as stacked / block-diagonal stages (hip.stacked_linear: layer 1 one
    128 -> 384 stage, layers 2 / 3 groups = 3, the 2-wide outputs padded to 16) == nine per-branch stages, bit for bit; the
    stacked copies follow in-place weight updates."""
    M = import_module("trafficbots_amd.models.modules")
    m, _ = _filled(tb, M.action_head.ActionHead, 21, dev, hidden_dim=128, action_dim=2, n_layer=3, mlp_use_layernorm=False,
                   log_std=-2.0, branch_type=True)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 37, 128, generator=g).to(dev)
    ty = torch.nn.functional.one_hot(torch.randint(0, 3, (3, 37), generator=g), 3).bool().to(dev)
    valid = (torch.rand(3, 37, generator=g) > 0.2).to(dev)
    outs = []
    for fused in (True, False, True):
        m.fused_branches = fused
        outs.append(m(x, valid, ty).mean.clone())
        if len(outs) == 2:
            with torch.no_grad():  # an optimizer-like in-place update: the stacked weights must be rebuilt
                for p in m.parameters():
                    p.mul_(1.01)
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[2], outs[0])
    m.fused_branches = False
    assert torch.equal(m(x, valid, ty).mean, outs[2])
    # the masked sum over the branches in the storing stage (TBX_F_MASKED_SUM) == ROWMASK + COPY / ADD per branch + STORE
    m.fused_branches, m.masked_sum_store = True, False
    assert torch.equal(m(x, valid, ty).mean, outs[2])
    assert bool((~valid).any()) and float(m(x, valid, ty).mean[~valid].abs().max()) == 0.0


@pytest.mark.parametrize("live", [1, 2, 4])
@pytest.mark.parametrize("rows", [1, 7, 64, 130])
def test_live_row_chain_is_bit_identical_to_mfma_tiles(tb, live, rows):
    """tbx_rowchain_live (tiles of 1 / 2 / 4 rows, LINEAR = a thread per output column running the MFMA sequence's k order as a
    v_fma chain) vs the 16-row MFMA tiles on the same program: every stage kind the transformer-layer and head chains use -
    LN, 128->128 / 128->384 / 128->512 / 512->128, block-diagonal per-head folds in both orientations, accumulate + row skip into
    the residual, odd widths (20 -> 64 -> 2), straight-to-global outputs. Bit for bit."""
    hip = import_module("trafficbots_amd.hip")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 10 + live)
    R = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
    d = 128
    x, z20 = R(rows, d), R(rows, 20)
    ln_w, ln_b = R(d) + 1.0, R(d)
    w_in, b_in = R(3 * d, d), R(3 * d)
    w_rpe, b_rpe = R(2 * d, d), R(2 * d)
    w_o, b_o = R(d, d), R(d)
    w1, b1, w2, b2 = R(4 * d, d), R(4 * d), R(d, 4 * d), R(d)
    wa, ba, wb, bb = R(64, 20), R(64), R(2, 64), R(2)
    skip = (torch.rand(rows, generator=g) < 0.3).to(torch.uint8).to(dev)
    outs = {}
    for mode in (0, live):
        ch = hip.Chain(16, 1028, 132, 132, live_rows=mode) if mode else hip.Chain(16, 1028)
        o_qkv, o_x, o_kv, o_small = (torch.zeros(rows, n, device=dev) for n in (7 * d, d, 2 * d, 2))
        ch.load(x, hip.BUF1, 0, n=d)
        ch.layernorm(hip.BUF1, 0, hip.BUF0, 0, ln_w, ln_b, 1e-5)
        ch.linear(hip.BUF0, 0, hip.BUF0, d, w_in, b_in)                                   # q | k | v
        ch.linear(hip.BUF0, d, hip.BUF0, 4 * d, w_rpe[:d], wt=True, groups=4, src_stride=32, dst_stride=d)  # qt = per-head W_k^T q
        ch.store(hip.BUF0, d, 7 * d, o_qkv)
        # per head: (k-slot)_h += W_rpe_v,h qt_h + b  (block-diagonal, accumulate), then x += skip ? 0 : out_proj(.)
        ch.linear(hip.BUF0, 4 * d, hip.BUF0, 2 * d, w_rpe[d:], b_rpe[d:], accum=True, groups=4, src_stride=d, dst_stride=32)
        ch.linear(hip.BUF0, 2 * d, hip.BUF1, 0, w_o, b_o, accum=True, skip_rows=skip)
        ch.layernorm(hip.BUF1, 0, hip.BUF0, 0, ln_w, ln_b, 1e-5)
        ch.linear(hip.BUF0, 0, hip.BUF0, d, w1, b1, relu=True)
        ch.linear(hip.BUF0, d, hip.BUF1, 0, w2, b2, accum=True)
        ch.rowmask(hip.BUF1, 0, d, mask=skip)
        ch.store(hip.BUF1, 0, d, o_x)
        ch.linear(hip.BUF1, 0, hip.GLOBAL, 0, w_in[d:], b_in[d:], out=o_kv)               # straight to global
        ch.load(z20, hip.BUF0, 0, n=20, pad_to=32)
        ch.linear(hip.BUF0, 0, hip.BUF0, 64, wa, ba, relu=True)
        ch.linear(hip.BUF0, 64, hip.AUX, 0, wb, bb)
        ch.clamp(hip.AUX, 0, 2, -0.25, 0.25)
        ch.store(hip.AUX, 0, 2, o_small)
        ch.run(rows)
        outs[mode] = (o_qkv, o_x, o_kv, o_small)
    torch.cuda.synchronize()
    for a, b, name in zip(outs[0], outs[live], ("qkv|qt", "x", "kv (global)", "small")):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("S", [33, 1030])  # 1030: n * S >= 1024 rows -> the wave-per-row form (4 rows per workgroup, ragged tail)
@pytest.mark.parametrize("bf16", [False, True])
def test_folded_attention_epilogue_equals_the_fold_stage(tb, bf16, S):
    """tbx_knarpe_attn_fwd_folded (the value half of linear_rpe applied in the attention kernel's epilogue, 128 floats per row
    out) vs tbx_knarpe_attn_fwd's 640-wide row followed by the grouped LINEAR stage that applied the fold so far: bit-identical
    (same fma order), rows without a valid target flagged the same - for a one-segment and a two-segment call."""
    hip = import_module("trafficbots_amd.hip")
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    n, T, K, d = 2, 60, 13, 128
    att = M.attention_rpe.AttentionRPE(d_model=d, n_head=4, dropout_p=0.0, d_rpe=d)
    tb.utils.det_fill(att, 15)
    att = att.to(dev)
    kv = torch.randn(n * T, 256, generator=g).to(dev)
    kv2 = torch.randn(n * 40, 256, generator=g).to(dev)
    if bf16:
        kv, kv2 = kv.to(torch.bfloat16), kv2.to(torch.bfloat16)
    q = torch.randn(n * S, 640, generator=g).to(dev)
    mk = lambda T_, K_: (torch.randint(0, T_, (n, S, K_), generator=g).to(torch.int32).to(dev),
                         (torch.rand(n, S, K_, generator=g) < 0.3).to(torch.uint8).to(dev), torch.randn(n, S, K_, d, generator=g).to(dev))
    (i1, m1, e1), (i2, m2, e2) = mk(T, K), mk(40, 7)
    m1[0, 2], m2[0, 2] = 1, 1  # a row without any valid target
    for segs in ([hip.Seg(kv, 0, 128, T, i1, m1, e1)], [hip.Seg(kv, 0, 128, T, i1, m1, e1), hip.Seg(kv2, 0, 128, 40, i2, m2, e2)]):
        o640, f640 = torch.empty(n * S, 640, device=dev), torch.empty(n * S, dtype=torch.uint8, device=dev)
        hip.knarpe_attn(q, 0, 128, att.linear_rpe.bias, n, S, segs, o640, f640)
        want = torch.empty(n * S, d, device=dev)
        ch = hip.Chain(16, 644)
        ch.load(o640, hip.BUF0, 0, n=640)
        ch.linear(hip.BUF0, d, hip.BUF0, 0, att.linear_rpe.weight[d:], att.linear_rpe.bias[d:], accum=True, groups=4, src_stride=d, dst_stride=32)
        ch.store(hip.BUF0, 0, d, want)
        ch.run(n * S)
        o128, f128 = torch.full((n * S, d), 7.0, device=dev), torch.empty(n * S, dtype=torch.uint8, device=dev)
        hip.knarpe_attn(q, 0, 128, att.linear_rpe.bias, n, S, segs, o128, f128, fold=eng.attn_fold_image(att))
        assert torch.equal(f128, f640) and int(f640.sum()) >= 1
        assert torch.equal(o128, want), float((o128 - want).abs().max())


@pytest.mark.parametrize("S,Ks,T,K,n_cross,bf16", [(8, 4, 64, 8, 1, False), (64, 25, 1024, 64, 1, False), (64, 36, 128, 24, 2, False),
                                                  (33, 13, 60, 128, 1, True)])
def test_fused_decoder_mid_launch_equals_its_three_launches(tb, hip, dev, S, Ks, T, K, n_cross, bf16):
    """tbx_knarpe_dec_mid (self attention -> x += out_proj -> LN -> q -> W_k^T q -> cross attention, one launch per decoder
    layer, csrc/dec_mid.hip) vs the three launches it replaces (folded attention kernel -> row chain -> folded attention
    kernel; transformer_rpe.py:165-192 of the reference): x after the self-attention residual, the cross attention's folded
    output and its no-valid-target flags are bit-identical - one- and two-segment cross sets, fp32 and bf16 K/V tables, rows
    whose self or cross set has no valid target."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    P = import_module("trafficbots_amd.utils.pose_emb")
    Seg, BUF1, D = hip.Seg, hip.BUF1, 128
    g = torch.Generator().manual_seed(S * 131 + K)
    blk = M.TransformerBlockRPE(n_layer=1, mode="dec_cross_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 5)
    layer = blk.to(dev).eval().layers[0]
    a1, a2 = layer.attn_src, layer.attn
    n, rows = 2, 2 * S
    x0 = torch.randn(rows, D, generator=g).to(dev)
    qkv = torch.randn(rows, 896, generator=g).to(dev)

    def knn(T_, K_):
        rel = torch.cat([(torch.rand(n, S, K_, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K_, 1, generator=g) - 0.5) * 6], -1)
        return (torch.randint(0, T_, (n, S, K_), generator=g).to(torch.int32).to(dev),
                (torch.rand(n, S, K_, generator=g) < 0.3).to(torch.uint8).to(dev), rel.to(dev).contiguous())

    i0, m0, r0 = knn(S, Ks)
    m0[0, 1] = 1  # a row without a valid self target: x keeps its value (attention_rpe.py:151-156 zero row -> out_proj skipped)
    kv16 = None
    if bf16:
        kv16 = qkv[:, D:3 * D].to(torch.bfloat16).contiguous()
    self_seg = Seg(qkv, D, 2 * D, S, i0, m0, None, rel=r0) if kv16 is None else Seg(kv16, 0, D, S, i0, m0, None, rel=r0)
    cross = []
    for c in range(n_cross):
        Tc, Kc = (T, K) if c == 0 else (40, 7)
        kv = torch.randn(n * Tc, 256, generator=g).to(dev)
        ic, mc, rc = knn(Tc, Kc)
        mc[1, 2] = 1  # ... and one without a valid cross target
        cross.append(Seg(kv.to(torch.bfloat16) if bf16 else kv, 0, D, Tc, ic, mc, None, 1, rel=rc))
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    fxy, fyw = pe.pe_xy.freqs, pe.pe_yaw.freqs
    # ---- the three launches
    ob1, f1 = torch.empty(rows, D, device=dev), torch.empty(rows, dtype=torch.uint8, device=dev)
    hip.knarpe_attn(qkv, 0, 3 * D, a1.linear_rpe.bias, n, S, [self_seg], ob1, f1, fxy, fyw, fold=eng.attn_fold_image(a1))
    x_want, q2 = x0.clone(), torch.empty(rows, 640, device=dev)
    ch = eng.layer_chain(rows)
    eng.emit_attn_out(ch, a1, ob1, f1, x=x_want)
    ch.store(BUF1, 0, D, x_want)
    eng.emit_proj(ch, rows, layer.norm1, a2, q2, with_kv=False)
    ch.run(rows)
    o_want, f_want = torch.empty(rows, D, device=dev), torch.empty(rows, dtype=torch.uint8, device=dev)
    hip.knarpe_attn(q2, 0, D, a2.linear_rpe.bias, n, S, cross, o_want, f_want, fxy, fyw, fold=eng.attn_fold_image(a2))
    # ---- one launch
    x_got, o_got, f_got = x0.clone(), torch.full((rows, D), 7.0, device=dev), torch.empty(rows, dtype=torch.uint8, device=dev)
    hip.knarpe_dec_mid(qkv, 0, 3 * D, x_got, self_seg, cross, a1.linear_rpe.bias, a2.linear_rpe.bias,
                       (layer.norm1.weight, layer.norm1.bias, layer.norm1.eps), n, S, eng.attn_fold_image(a1),
                       hip.packed_weight(a1.out_proj_weight, a1.out_proj_bias, gemv=True),
                       hip.packed_weight(a2.in_proj_weight[:D], a2.in_proj_bias[:D], gemv=True),
                       hip.packed_weight(a2.linear_rpe.weight[:D], None, wt=True, groups=4, gemv=True), eng.attn_fold_image(a2),
                       o_got, f_got, fxy, fyw)
    torch.cuda.synchronize()
    assert int(f1.sum()) >= 1 and int(f_want.sum()) >= 1
    assert torch.equal(f_got, f_want)
    assert torch.equal(x_got, x_want), float((x_got - x_want).abs().max())
    assert torch.equal(o_got, o_want), float((o_got - o_want).abs().max())


def test_knn_multi_launch_equals_single_searches(hip, dev):
    """tbx_knn_embed_multi (the agents' three searches of a step in one launch) vs three tbx_knn_embed calls: identical index
    sets, masks, relative poses and embeddings (the same device code per (job, row))."""
    g = torch.Generator().manual_seed(77)
    n, S = 2, 37
    pose, inv = _poses(g, n, S).to(dev), (torch.rand(n, S, generator=g) < 0.2).to(torch.uint8).to(dev)
    tg = [(_poses(g, 1, 1024).to(dev), (torch.rand(1, 1024, generator=g) < 0.3).to(torch.uint8).to(dev), 64, 2),
          (pose, inv, 12, 1), (_poses(g, n, 40).to(dev), (torch.rand(n, 40, generator=g) < 0.3).to(torch.uint8).to(dev), 8, 1),
          (_poses(g, n, 1500).to(dev), torch.zeros(n, 1500, dtype=torch.uint8, device=dev), 33, 1)]
    P = import_module("trafficbots_amd.utils.pose_emb")
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    for want_emb in (False, True):
        jobs = [dict(src_pose=pose, src_invalid=inv, tgt_pose=tp, tgt_invalid=ti, k=k, dist_limit=150.0, tgt_batch_div=div,
                     want_rel_pose=True, want_emb=want_emb) for tp, ti, k, div in tg]
        got = hip.knn_embed_multi(jobs, pe.pe_xy.freqs, pe.pe_yaw.freqs)
        for q, o in zip(jobs, got):
            want = hip.knn_embed(freqs_xy=pe.pe_xy.freqs, freqs_yaw=pe.pe_yaw.freqs, **q)
            for a, b in zip(o, want):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b)


@pytest.mark.parametrize("S,Ks,T,K,n_layer,bf16", [(8, 4, 64, 8, 2, False), (64, 25, 1024, 64, 3, False), (33, 13, 60, 40, 2, True)])
def test_one_launch_decoder_layer_equals_mid_launch_plus_chain(tb, hip, dev, S, Ks, T, K, n_layer, bf16):
    """tbx_knarpe_dec_layer (a whole dec_cross_attn layer per launch: attention half, out_proj, FFN, row mask, the next layer's
    q | k | v | W_k^T q; transformer_rpe.py:207-245) through engine.run_block vs the two launches per layer it replaces
    (tbx_knarpe_dec_mid + the row chain) and vs the 16-row MFMA schedule: the block's output rows are bit-identical, invalid
    source rows are 0, for 2 and 3 layers (the last layer has no projections to make)."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    P = import_module("trafficbots_amd.utils.pose_emb")
    g = torch.Generator().manual_seed(S * 17 + n_layer)
    blk = M.TransformerBlockRPE(n_layer=n_layer, mode="dec_cross_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 9)
    blk = blk.to(dev).eval()
    n, rows, D = 2, 2 * S, 128
    x0 = torch.randn(rows, D, generator=g).to(dev)
    src_invalid = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    x0[src_invalid.bool()] = 0.0

    def knn(T_, K_):
        rel = torch.cat([(torch.rand(n, S, K_, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K_, 1, generator=g) - 0.5) * 6], -1)
        return (torch.randint(0, T_, (n, S, K_), generator=g).to(torch.int32).to(dev),
                (torch.rand(n, S, K_, generator=g) < 0.3).to(torch.uint8).to(dev), rel.to(dev).contiguous())

    i0, m0, r0 = knn(S, Ks)
    m0[src_invalid.view(n, S).bool()] = 1  # an invalid source has no valid pair (as the K-nearest kernel produces)
    ic, mc, rc = knn(T, K)
    mc[src_invalid.view(n, S).bool()] = 1
    kv = torch.randn(n * T, n_layer * 256, generator=g).to(dev)
    if bf16:  # bfloat16 K/V tables (engine.KV_BF16): the one-launch layer also writes the next layer's bf16 k | v copy
        kv = kv.to(torch.bfloat16)
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    outs = {}
    # "layer_mf": the one-launch layer with its tail's LINEAR stages on the split-bf16 matrix path (Schedule.dec_tail_mfma, the default)
    for name, (live, fold, mid, layer, mf) in {"mfma": (0, False, False, False, False), "mid": (1, True, True, False, False),
                                               "layer": (1, True, True, True, False), "layer_mf": (1, True, True, True, True)}.items():
        x = x0.clone()
        with eng.use(eng.DEFAULT.replace(live_rows=live, attn_fold=fold, dec_mid=mid, dec_layer=layer, kv_bf16=bf16, split_bf16=False,
                                         dec_tail_mfma=mf)):
            eng.run_block(blk, x, src_invalid, n, S, eng.SelfKnn(i0, m0, rel=r0),
                          cross=lambda l: [hip.Seg(kv, l * 256, l * 256 + D, T, ic, mc, None, 1, rel=rc)], pose_rpe=pe)
        torch.cuda.synchronize()
        outs[name] = x
    assert torch.isfinite(outs["mfma"]).all() and float(outs["mfma"][src_invalid.bool()].abs().max()) == 0.0
    assert float((outs["mfma"] - x0).abs().max()) > 1e-3
    for name in ("mid", "layer"):
        assert torch.equal(outs[name], outs["mfma"]), (name, float((outs[name] - outs["mfma"]).abs().max()))
    # split-bf16 LINEAR stages: < 3e-5 of sum |x||w| per output; through 2-3 layers (LayerNorm, softmax over the changed q | k | v)
    # the rows stay within 2e-4 of the largest entry, and are not bit-identical (the matrix path did run)
    err = float((outs["layer_mf"] - outs["mfma"]).abs().max())
    assert 0.0 < err <= 2e-4 * float(outs["mfma"].abs().max()), err
    assert float(outs["layer_mf"][src_invalid.bool()].abs().max()) == 0.0


@pytest.mark.parametrize("tile,groups,gw", [(16, 7, 11), (32, 9, 11), (48, 10, 11), (16, 5, 16)])
def test_pooled_rows_kept_in_lds_feed_the_same_stages(hip, dev, tile, groups, gw):
    """TBX_F_POOL_KEEP: POOLMAX leaves the pooled rows in LDS and the tile goes on as a flat tile of its groups (several groups per
    32 / 48-row tile, a ragged last tile) - the stages after it give what a second launch over the pooled rows gives, bit for bit."""
    g = torch.Generator().manual_seed(tile + groups)
    x = torch.randn(groups * gw, 128, generator=g).to(dev)
    inv = (torch.rand(groups * gw, generator=g) < 0.3).to(torch.uint8)
    inv[:gw] = 1  # a group without a valid row -> pooled row of zeros
    inv = inv.to(dev)
    w, b = torch.randn(48, 128, generator=g).to(dev), torch.randn(48, generator=g).to(dev)
    lw, lb = torch.randn(128, generator=g).to(dev), torch.randn(128, generator=g).to(dev)
    C, B0, B1, AUX = hip.Chain, hip.BUF0, hip.BUF1, hip.AUX
    pooled_a, out_a = torch.empty(groups, 128, device=dev), torch.empty(groups, 48, device=dev)
    ch = C(tile, 260)
    ch.load(x, B0, 0, n=128)
    ch.poolmax(B0, 0, 128, pooled_a, mask=inv, keep=(AUX, 0))
    ch.layernorm(AUX, 0, B1, 0, lw, lb, 1e-5)
    ch.linear(B1, 0, B0, 0, w, b, relu=True)
    ch.store(B0, 0, 48, out_a)
    ch.run(groups * gw, group_rows=gw)
    pooled_b, out_b = torch.empty(groups, 128, device=dev), torch.empty(groups, 48, device=dev)
    ch = C(tile, 260)
    ch.load(x, B0, 0, n=128)
    ch.poolmax(B0, 0, 128, pooled_b, mask=inv)
    ch.run(groups * gw, group_rows=gw)
    ch = C(16, 260)
    ch.load(pooled_b, AUX, 0, n=128)
    ch.layernorm(AUX, 0, B1, 0, lw, lb, 1e-5)
    ch.linear(B1, 0, B0, 0, w, b, relu=True)
    ch.store(B0, 0, 48, out_b)
    ch.run(groups)
    assert torch.equal(pooled_a, pooled_b) and float(pooled_a[0].abs().max()) == 0.0
    assert torch.equal(out_a, out_b)
    assert torch.isfinite(out_a).all()


@pytest.mark.parametrize("mode,S,Ks,T,K,n_layer,bf16", [("dec_cross_attn", 600, 12, 300, 40, 2, False), ("enc_self_attn", 1031, 16, 0, 0, 3, False),
                                                       ("dec_cross_attn", 520, 9, 128, 24, 2, True)])
def test_layer_tile_equals_row_chains(tb, hip, dev, mode, S, Ks, T, K, n_layer, bf16):
    """tbx_layer_tile (large launches: a layer's row-local chains as straight-line 16-row tiles on the split-bf16 matrix path) through
    engine.run_block vs the exact-fp32 tbx_rowchain schedule, transformer_rpe.py:207-245. Tolerance: the split drops < 3e-5 of
    sum |x||w| per LINEAR output; over 2-3 layers the token rows agree to 2e-4 of their largest entry (the suite's fp32 tolerance).
    Invalid source rows are exactly 0 on both paths; the ragged last tile (rows % 16 != 0) is covered."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    P = import_module("trafficbots_amd.utils.pose_emb")
    g = torch.Generator().manual_seed(S + n_layer)
    blk = M.TransformerBlockRPE(n_layer=n_layer, mode=mode, d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 11)
    blk = blk.to(dev).eval()
    n, rows, D = 2, 2 * S, 128
    x0 = torch.randn(rows, D, generator=g).to(dev)
    src_invalid = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    x0[src_invalid.bool()] = 0.0

    def knn(T_, K_):
        rel = torch.cat([(torch.rand(n, S, K_, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K_, 1, generator=g) - 0.5) * 6], -1)
        m = (torch.rand(n, S, K_, generator=g) < 0.3).to(torch.uint8).to(dev)
        m[src_invalid.view(n, S).bool()] = 1
        m[0, 3] = 1  # a valid source without any valid target: its attention update is skipped
        return torch.randint(0, T_, (n, S, K_), generator=g).to(torch.int32).to(dev), m, rel.to(dev).contiguous()

    i0, m0, r0 = knn(S, Ks)
    cross = None
    if mode == "dec_cross_attn":
        ic, mc, rc = knn(T, K)
        kv = torch.randn(n * T, n_layer * 256, generator=g).to(dev)
        if bf16:
            kv = kv.to(torch.bfloat16)
        cross = lambda l: [hip.Seg(kv, l * 256, l * 256 + D, T, ic, mc, None, 1, rel=rc)]
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    outs = {}
    for name, tile in (("chain", False), ("tile", True)):
        x = x0.clone()
        with eng.use(eng.DEFAULT.replace(tile_layer=tile, tile_min_rows=1024, kv_bf16=bf16, attn_fold_big=False)):
            assert eng.tile_rows_ok(rows) == tile
            eng.run_block(blk, x, src_invalid, n, S, eng.SelfKnn(i0, m0, rel=r0), cross=cross, pose_rpe=pe)
        torch.cuda.synchronize()
        outs[name] = x
    ref, got = outs["chain"], outs["tile"]
    assert torch.isfinite(ref).all() and float((ref - x0).abs().max()) > 1e-3
    assert float(got[src_invalid.bool()].abs().max()) == 0.0 and float(ref[src_invalid.bool()].abs().max()) == 0.0
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= (2e-4 if not bf16 else 2e-3) * scale, (err, scale)  # (bf16 tables: the k | v rows are rounded to bf16 AFTER a slightly different fp32 value)
    assert err > 0.0  # the two paths really are different arithmetic


@pytest.mark.parametrize("bf16_products", [False, True])
def test_layer_tile_two_workgroups_per_cu_form_is_bit_identical(tb, hip, dev, bf16_products):
    """tile_layer.hip picks the depth of its weight ring by the launch's size: at least one tile per CU -> two register slots (<= 128
    VGPRs: two workgroups per CU), fewer -> three. Same stages, same operands, same order: a block run on 2 x 2,100 rows in one call
    (263 tiles) must give the bits of the two batch entries run one at a time (132 tiles each), for every instantiation a decoder block
    launches (<0,0,2> first projection, <1,0,1>, <1,1,2>, <1,1,0>). transformer_rpe.py:207-245."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    P = import_module("trafficbots_amd.utils.pose_emb")
    g = torch.Generator().manual_seed(77)
    blk = M.TransformerBlockRPE(n_layer=2, mode="dec_cross_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 13)
    blk = blk.to(dev).eval()
    n, S, Ks, T, K, D = 2, 2100, 8, 96, 12, 128
    x0 = torch.randn(n * S, D, generator=g).to(dev)
    src_invalid = (torch.rand(n * S, generator=g) < 0.1).to(torch.uint8).to(dev)
    x0[src_invalid.bool()] = 0.0

    def knn(T_, K_):
        rel = torch.cat([(torch.rand(n, S, K_, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K_, 1, generator=g) - 0.5) * 6], -1)
        m = (torch.rand(n, S, K_, generator=g) < 0.3).to(torch.uint8).to(dev)
        m[src_invalid.view(n, S).bool()] = 1
        return torch.randint(0, T_, (n, S, K_), generator=g).to(torch.int32).to(dev), m, rel.to(dev).contiguous()

    i0, m0, r0 = knn(S, Ks)
    ic, mc, rc = knn(T, K)
    kv = torch.randn(n * T, 2 * 256, generator=g).to(dev)
    pe = P.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3).to(dev)
    sched = eng.DEFAULT.replace(tile_layer=True, tile_min_rows=1024, attn_fold_big=False, linear_bf16=bf16_products)

    def run(b0, b1):
        nb = b1 - b0
        x = x0[b0 * S:b1 * S].clone()
        cross = lambda l: [hip.Seg(kv[b0 * T:b1 * T].contiguous(), l * 256, l * 256 + D, T, ic[b0:b1].contiguous(), mc[b0:b1].contiguous(), None, 1,
                                   rel=rc[b0:b1].contiguous())]
        with eng.use(sched):
            assert eng.tile_rows_ok(nb * S)
            eng.run_block(blk, x, src_invalid[b0 * S:b1 * S].contiguous(), nb, S,
                          eng.SelfKnn(i0[b0:b1].contiguous(), m0[b0:b1].contiguous(), rel=r0[b0:b1].contiguous()), cross=cross, pose_rpe=pe)
        torch.cuda.synchronize()
        return x

    whole = run(0, 2)
    halves = torch.cat([run(0, 1), run(1, 2)], 0)
    assert torch.isfinite(whole).all() and float((whole - x0).abs().max()) > 1e-3
    assert torch.equal(whole, halves)


@pytest.mark.parametrize("rows,r_rows,pose3", [(64, 64, False), (40, 21, False), (16, 130, True), (48, 33, True)])
def test_layer_tile_rider_equals_the_navigation_chain(tb, hip, dev, rows, r_rows, pose3):
    """tbx_layer_tile_t's rider (extra workgroups of the first-projection launch): y = add + W0 in + b0, three relu LINEARs, invalid
    rows 0 - against the row chain the engine runs for the heads' navigation embedding (navigation.py:65-79 +
    add_navi_latent.py:43-50: LINEAR accumulate, three LINEAR + relu, the last with its row mask), on ragged row counts that differ
    from the main rows'; the main rows' projections are the same bits with and without the rider. pose3: the stage-0 input is the
    pose embedding the rider builds itself from [rows, 3] poses (large launches) - the chain reads tbx_pose_embed's."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    g = torch.Generator().manual_seed(rows + r_rows)
    blk = M.TransformerBlockRPE(n_layer=1, mode="enc_self_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.0,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 5)
    blk = blk.to(dev).eval()
    l0 = blk.layers[0]
    D = 128
    x = torch.randn(rows, D, generator=g).to(dev)
    Ws = [(torch.randn(D, D, generator=g) / 8).to(dev) for _ in range(4)]
    bs = [torch.randn(D, generator=g).to(dev) for _ in range(4)]
    inp, add = torch.randn(r_rows, D, generator=g).to(dev), torch.randn(r_rows, D, generator=g).to(dev)
    src = dict(inp=inp)
    if pose3:
        p3 = torch.cat([(torch.rand(r_rows, 2, generator=g) - 0.5) * 300, (torch.rand(r_rows, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
        inp = hip.pose_embed(p3, fxy, fyw, D)
        src = dict(pose3=p3, freqs=(fxy, fyw))
    valid = (torch.rand(r_rows, generator=g) < 0.7).to(torch.uint8).to(dev)
    out = torch.full((r_rows, D), 7.0, device=dev)
    qkv = [torch.zeros(rows, eng.QKV_LD, device=dev) for _ in range(2)]
    with eng.use(eng.DEFAULT):
        proj = lambda o: eng.tile_proj_part(l0.norm1, l0.attn, o, True, None)
        hip.layer_tile(x, proj=proj(qkv[0]), store_x=False)
        hip.layer_tile(x, proj=proj(qkv[1]), store_x=False,
                       rider=dict(add=add, out=out, valid=valid, images=[hip.packed_weight(w, b, mfma32=True) for w, b in zip(Ws, bs)], **src))
    ref = torch.empty(r_rows, D, device=dev)
    C, B0 = hip.Chain, hip.BUF0
    ch = C(16, 4 * D + 4)
    ch.load2(inp, B0, 0, add, B0, D)
    ch.linear(B0, 0, B0, D, Ws[0], bs[0], accum=True)
    ch.linear(B0, D, B0, 2 * D, Ws[1], bs[1], relu=True)
    ch.linear(B0, 2 * D, B0, D, Ws[2], bs[2], relu=True)
    ch.linear(B0, D, B0, 2 * D, Ws[3], bs[3], relu=True, skip_rows=valid, skip_is_valid=True, zero_skipped=True)
    ch.store(B0, 2 * D, D, ref)
    ch.run(r_rows)
    torch.cuda.synchronize()
    assert torch.equal(qkv[0], qkv[1]) and float(qkv[0].abs().max()) > 0
    assert float(out[~valid.bool()].abs().max()) == 0.0 and float(ref[~valid.bool()].abs().max()) == 0.0
    err, scale = float((out - ref).abs().max()), float(ref.abs().max())
    assert 0.0 < err <= 1e-4 * scale, (err, scale)


@pytest.mark.parametrize("n,k,groups,wt", [(128, 128, 1, False), (512, 128, 1, False), (128, 512, 1, False), (384, 128, 1, False),
                                           (32, 128, 4, False), (128, 32, 4, True), (64, 64, 1, False), (64, 32, 1, False), (16, 128, 3, False)])
def test_pack_weight_mfma32_layout(hip, dev, n, k, groups, wt):
    """The tbx_pack_weight_mfma32 image, decoded on the host by the layout rule of csrc/tile_layer.hip (units of 4 groups of one
    (16-channel tile, 32-k step) each + the groups' tile biases), gives back bf16 hi + lo halves with hi + lo == w to 2^-15 relative,
    every (output channel, k) exactly once, and each group's bias."""
    g = torch.Generator().manual_seed(n + k)
    w = (torch.randn(groups * (k if wt else n), n if wt else k, generator=g)).to(dev)
    bias = torch.randn(groups * n, generator=g).to(dev)
    img = hip.packed_weight(w, bias, wt=wt, groups=groups, mfma32=True).cpu()
    U = 2112
    units = img.numel() // U
    T = groups * n // 16
    Wd = torch.zeros(groups * n, k)
    seen = torch.zeros(groups * n, k)
    wc, bc = w.cpu(), bias.cpu()
    for u in range(units):
        blk = img[u * U:(u + 1) * U]
        raw = blk[:2048].view(torch.int32).view(4, 2, 64, 4)  # group, hi / lo, lane, dword
        el = torch.stack([raw & 0xFFFF, (raw >> 16) & 0xFFFF], -1).reshape(4, 2, 64, 8)  # 8 bf16 bit patterns per lane
        val = (el << 16).view(torch.float32)
        full = val[:, 0] + val[:, 1]  # hi + lo: [group, lane, 8]
        for s in range(4):
            if k == 32:
                tile, step = u * 4 + s, 0
            elif k == 64:
                tile, step = u * 2 + (s >> 1), s & 1
            else:
                tile, step = u % T, 4 * (u // T) + s
            if tile >= T:
                assert float(full[s].abs().max()) == 0.0
                continue
            for l in range(64):
                oc = tile * 16 + (l & 15)
                kk = step * 32 + (l >> 4) * 8
                Wd[oc, kk:kk + 8] = full[s, l]
                seen[oc, kk:kk + 8] += 1
            assert torch.equal(blk[2048 + 16 * s:2048 + 16 * s + 16], bc[tile * 16:tile * 16 + 16])
    assert bool((seen == 1).all())
    ref = torch.zeros(groups * n, k)
    for grp in range(groups):
        ref[grp * n:(grp + 1) * n] = wc[grp * k:(grp + 1) * k].t() if wt else wc[grp * n:(grp + 1) * n]
    assert float((Wd - ref).abs().max()) <= 2.0 ** -15 * float(ref.abs().max())


def _default_model(tb, dev):
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 3)
    return wm.to(dev).eval()


@pytest.mark.parametrize("G,W", [(1031, 11), (200, 11), (64, 7)])
def test_window_tile_equals_grouped_chain(tb, hip, dev, G, W):
    """tbx_window_tile (the agents' temporal PointNet as one launch: input MLP, 3 PointNet layers with the window maximum taken across
    the lanes of a DPP row, pooled row; agent_encoder.py:130-159, polyline_encoder.py:49-61) vs the grouped tbx_rowchain program it
    replaces for large launches: 2e-4 of the largest pooled entry (split-bf16 stages vs exact fp32); windows without a valid row
    are exactly 0; a ragged last workgroup (odd G) and short windows (W = 7) are covered."""
    eng = import_module("trafficbots_amd.engine")
    ae = _default_model(tb, dev).model.ag_encoder
    g = torch.Generator().manual_seed(G)
    attr = torch.zeros(G * W, 32)
    attr[:, :20] = torch.randn(G * W, 20, generator=g)
    pe = torch.randn(G * W, 64, generator=g)
    inv = (torch.rand(G, W, generator=g) < 0.3)
    inv[5] = True  # a window without a valid row
    inv[7, 1:] = True
    attr, pe, inv8 = attr.to(dev), pe.to(dev), inv.reshape(-1).to(torch.uint8).to(dev)
    ref = torch.empty(G, 128, device=dev)
    ch = hip.Chain(hip.group_tile_rows(W, G), 132)
    ch.load2(attr, hip.BUF0, 0, pe, hip.BUF1, 64)
    cur = eng.emit_mlp(ch, ae.input_encoder.mlp, hip.BUF0, 0)
    eng.emit_pointnet(ch, ae.temp_encoder, inv8, ref, x_buf=cur)
    ch.run(G * W, group_rows=W)
    imgs = ae._window_tile_images(32)
    assert imgs is not None
    out = torch.full((G, 128), 7.0, device=dev)
    hip.window_tile(attr, pe, inv8, imgs[0], imgs[1], W, out)
    torch.cuda.synchronize()
    assert float(out[5].abs().max()) == 0.0 and float(ref[5].abs().max()) == 0.0
    scale = float(ref.abs().max())
    assert scale > 1e-2
    assert float((out - ref).abs().max()) <= 2e-4 * scale, (float((out - ref).abs().max()), scale)


@pytest.mark.parametrize("G,W,add,rider_rows,knn", [(64, 11, False, 64, True), (37, 11, False, 21, True), (128, 11, True, 0, False), (5, 7, False, 0, True)])
def test_front_equals_its_three_launches(tb, hip, dev, G, W, add, rider_rows, knn):
    """tbx_front - window PointNet + the first projection of its pooled rows (+ rider) + K-nearest searches and pose-embedding job in
    ONE launch (csrc/front.hip) - against tbx_window_tile -> tbx_layer_tile (rider) and tbx_knn_embed_multi_pe as launches of their
    own: the same device functions on the same operands, so pooled rows, q | k | v | W_k^T q rows, rider rows, K-nearest indices /
    masks / relative poses and the embedded poses are bit-identical; odd window counts (a ragged last workgroup), a rider whose row
    count differs from the windows', the lights' "add" mode without searches, a launch smaller than one rider tile."""
    eng = import_module("trafficbots_amd.engine")
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    wm = _default_model(tb, dev)
    enc = wm.model.tl_encoder if add else wm.model.ag_encoder
    g = torch.Generator().manual_seed(G * 3 + W)
    D = 128
    if add:
        attr = torch.randn(G * W, 16, generator=g).to(dev)
        pe = torch.randn(G, D, generator=g).to(dev)
        imgs = enc._window_tile_images()
    else:
        attr = torch.zeros(G * W, 32)
        attr[:, :20] = torch.randn(G * W, 20, generator=g)
        attr, pe = attr.to(dev), torch.randn(G * W, 64, generator=g).to(dev)
        imgs = enc._window_tile_images(32)
    inv = torch.rand(G, W, generator=g) < 0.3
    inv[min(3, G - 1)] = True
    inv8 = inv.reshape(-1).to(torch.uint8).to(dev)
    blk = M.TransformerBlockRPE(n_layer=1, mode="enc_self_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.0,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, 6)
    l0 = blk.to(dev).eval().layers[0]
    rider = lambda out: None
    if rider_rows:
        Ws = [(torch.randn(D, D, generator=g) / 8).to(dev) for _ in range(4)]
        bs = [torch.randn(D, generator=g).to(dev) for _ in range(4)]
        p3 = torch.cat([(torch.rand(rider_rows, 2, generator=g) - 0.5) * 300, (torch.rand(rider_rows, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
        add_rows = torch.randn(rider_rows, D, generator=g).to(dev)
        valid = (torch.rand(rider_rows, generator=g) < 0.7).to(torch.uint8).to(dev)
        rider = lambda out: dict(pose3=p3, freqs=(fxy, fyw), add=add_rows, out=out, valid=valid,
                                 images=[hip.packed_weight(w, b, mfma32=True) for w, b in zip(Ws, bs)])
    jobs = lambda: None
    if knn:
        S, T = G, 200
        sp = torch.cat([(torch.rand(1, S, 2, generator=g) - 0.5) * 200, (torch.rand(1, S, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        tp = torch.cat([(torch.rand(1, T, 2, generator=g) - 0.5) * 200, (torch.rand(1, T, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        si, ti = (torch.rand(1, S, generator=g) < 0.1).to(torch.uint8).to(dev), (torch.rand(1, T, generator=g) < 0.2).to(torch.uint8).to(dev)
        common = dict(src_pose=sp, src_invalid=si, dist_limit=150.0, want_rel_pose=True, want_emb=False)
        jobs = lambda: [dict(common, tgt_pose=tp, tgt_invalid=ti, k=16), dict(common, tgt_pose=sp, tgt_invalid=si, k=min(4, S - 1))]
    pose_job = lambda out: None
    if knn and rider_rows == 0:  # (the searches' pose-embedding job: not together with a rider that embeds poses itself)
        q3 = torch.randn(G, 3, generator=g).to(dev)
        fxy2, fyw2 = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)
        pose_job = lambda out: dict(pose3=q3, freqs_xy=fxy2, freqs_yaw=fyw2, pe_dim=128, out=out)
    res = {}
    for fused in (False, True):
        x = torch.full((G, D), 7.0, device=dev)
        qkv = torch.zeros(G, eng.QKV_LD, device=dev)
        r_out = torch.full((max(rider_rows, 1), D), 5.0, device=dev)
        pj_out = torch.full((G, D), 3.0, device=dev)
        with eng.use(eng.DEFAULT):
            win = dict(attr=attr, pe=pe, row_invalid=inv8, in_images=imgs[0], pn_images=imgs[1], window=W, out=x, add_mode=add)
            proj = eng.tile_proj_part(l0.norm1, l0.attn, qkv, True, None)
            if fused:
                outs = hip.front(window=win, proj=proj, rider=rider(r_out), jobs=jobs(), pose_embed_job=pose_job(pj_out))
            else:
                hip.window_tile(**win)
                hip.layer_tile(x, proj=proj, store_x=False, rider=rider(r_out))
                outs = hip.knn_embed_multi(jobs(), pose_embed_job=pose_job(pj_out)) if knn else []
        torch.cuda.synchronize()
        res[fused] = (x, qkv, r_out, pj_out, outs)
    a, b = res[False], res[True]
    assert float(a[0].abs().max()) > 1e-2 and float(a[0][min(3, G - 1)].abs().max()) == 0.0  # (a window without a valid row)
    for i, name in enumerate(("pooled rows", "q | k | v | qt", "rider rows", "embedded poses")):
        assert torch.equal(a[i], b[i]), name
    assert len(a[4]) == len(b[4])
    for (i0, m0, r0, _), (i1, m1, r1, _) in zip(a[4], b[4]):
        assert torch.equal(i0, i1) and torch.equal(m0, m1) and torch.equal(r0, r1)


@pytest.mark.parametrize("G,W", [(300, 11), (128, 11), (33, 5)])
def test_window_tile_add_mode_equals_grouped_chain(tb, hip, dev, G, W):
    """tbx_window_tile in "add" mode = the traffic lights' temporal encoder (traffic_light.py:219-226: input MLP 16 -> 128 -> 128 ->
    128 on the one-hot state / window rows, + the light's lane feature, then the 3 PointNet layers and the pooled row) vs the
    grouped tbx_rowchain program: 2e-4 of the largest entry; 16-column attribute rows (the kernel reads the 32-wide first layer's
    missing columns as zeros)."""
    eng = import_module("trafficbots_amd.engine")
    te = _default_model(tb, dev).model.tl_encoder
    g = torch.Generator().manual_seed(G + W)
    attr = torch.randn(G * W, 16, generator=g).to(dev)
    feat = torch.randn(G, 128, generator=g).to(dev)
    inv = (torch.rand(G, W, generator=g) < 0.3)
    inv[2] = True
    inv8 = inv.reshape(-1).to(torch.uint8).to(dev)
    ref = torch.empty(G, 128, device=dev)
    ch = hip.Chain(hip.group_tile_rows(W, G), 132)
    cur = te.input_encoder.emit(ch, attr, feat, pe_row_div=W)
    eng.emit_pointnet(ch, te.temp_encoder, inv8, ref, x_buf=cur)
    ch.run(G * W, group_rows=W)
    imgs = te._window_tile_images()
    assert imgs is not None
    out = torch.full((G, 128), 7.0, device=dev)
    hip.window_tile(attr, feat, inv8, imgs[0], imgs[1], W, out, add_mode=True)
    torch.cuda.synchronize()
    assert float(out[2].abs().max()) == 0.0 and float(ref[2].abs().max()) == 0.0
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 2e-4 * scale, (float((out - ref).abs().max()), scale)


@pytest.mark.parametrize("rows", [1030, 100])
def test_heads_tile_equals_heads_chain(tb, hip, dev, rows):
    """tbx_heads_tile (navigation / latent adders + the action head's stacked branches + masked sum in one launch,
    traffic_bots.py:206-221) vs the heads chain: actions within 2e-4 of the largest one; rows of no agent type sum to exactly 0."""
    eng = import_module("trafficbots_amd.engine")
    m = _default_model(tb, dev).model
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 128, generator=g).to(dev)
    navi_emb, lat_emb = torch.relu(torch.randn(rows, 128, generator=g)).to(dev), torch.relu(torch.randn(rows, 128, generator=g)).to(dev)
    navi_valid = (torch.rand(rows, generator=g) < 0.8).to(torch.uint8).to(dev)
    lat_inv = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    navi_emb[navi_valid == 0] = 0.0
    lat_emb[lat_inv != 0] = 0.0
    ty = torch.randint(0, 4, (rows,), generator=g)  # 3: no type (an invalid agent)
    type_mask = torch.stack([(ty != i) for i in range(3)]).to(torch.uint8).contiguous().to(dev)
    ref = torch.empty(rows, 2, device=dev)
    ch = hip.Chain(16, 4 * 128 + 4)
    ch.load(x, hip.BUF1, 0, n=128)
    m.add_navi.emit(ch, navi_valid, mask_is_valid=True, z_embedded=navi_emb, z_premasked=True)
    m.add_latent.emit(ch, lat_inv, None, z_embedded=lat_emb, z_premasked=True)
    m.action_head.emit(ch, type_mask, ref)
    ch.run(rows)
    pw = lambda w, b, **kw: hip.packed_weight(w, b, mfma32=True, **kw)
    lins = [[t[0] for t in mlp.linear_layers()] for mlp in m.action_head.mlp_mean]
    w1, b1 = hip.stacked_linear([l[0] for l in lins])
    w2, b2 = hip.stacked_linear([l[1] for l in lins])
    w3, b3 = hip.stacked_linear([l[2] for l in lins], pad_out_to=16)
    imgs = [pw(t[0].weight, t[0].bias) for t in m.add_navi.mlp.linear_layers()] + [pw(t[0].weight, t[0].bias) for t in m.add_latent.mlp.linear_layers()]
    imgs += [pw(w1, b1), pw(w2, b2, groups=3), pw(w3, b3, groups=3)]
    out = torch.full((rows, 2), 7.0, device=dev)
    hip.heads_tile(x, dict(images=imgs, navi_emb=navi_emb, latent_emb=lat_emb, navi_valid=navi_valid, latent_invalid=lat_inv,
                           type_mask=type_mask, action_out=out))
    torch.cuda.synchronize()
    none = (ty == 3).to(dev)
    assert float(out[none].abs().max()) == 0.0 and float(ref[none].abs().max()) == 0.0
    scale = float(ref.abs().max())
    assert scale > 1e-3
    assert float((out - ref).abs().max()) <= 2e-4 * scale, (float((out - ref).abs().max()), scale)


@pytest.mark.parametrize("rows,p", [(1040, 0.1), (77, 0.0)])
def test_heads_tile_raw_with_keyed_dropout_equals_the_two_chains(tb, hip, dev, rows, p):
    """tbx_heads_tile with raw = 1 (training's stepping pass): navigation embedding mlp_in(dest_feature + mlp_pe(pe)), latent embedding
    mlp_in(z), both adders with the keyed dropouts of their 12 relu outputs, action head (navigation.py:65-79, add_navi_latent.py:43-65,
    action_head.py:74-100) in ONE launch, against the row chains it replaces with the SAME dropout site ids (engine.DROP_CTX): the
    masks are tbx_keyed_dropout's in both (an element dropped in one is dropped in the other: any mismatch would be an O(1)
    difference), values within 2e-4 of the largest action; p = 0 (no dropout) and a ragged last tile are covered."""
    eng = import_module("trafficbots_amd.engine")
    m = _default_model(tb, dev).model
    for mod in (m.add_navi.mlp_in, m.add_navi.mlp, m.add_latent.mlp_in, m.add_latent.mlp):
        mod.dropout_p = p if p > 0 else None
    g = torch.Generator().manual_seed(rows)
    d = 128
    x = torch.randn(rows, d, generator=g).to(dev)
    navi_pe, dest_f = torch.randn(rows, d, generator=g).to(dev), torch.randn(rows, d, generator=g).to(dev)
    z = torch.randn(rows, 16, generator=g).to(dev)
    navi_valid = (torch.rand(rows, generator=g) < 0.8).to(torch.uint8).to(dev)
    lat_inv = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    ty = torch.randint(0, 4, (rows,), generator=g)
    type_mask = torch.stack([(ty != i) for i in range(3)]).to(torch.uint8).contiguous().to(dev)
    seed = torch.tensor([1234567], dtype=torch.int64, device=dev)
    outs = {}
    for name in ("chain", "tile"):
        eng.DROP_CTX = dict(seed=seed, site=40, call=3, step=7) if p > 0 else None
        try:
            out = torch.full((rows, 2), 7.0, device=dev)
            if name == "chain":
                ch = hip.Chain(16, 4 * d + 4)
                ch.load(x, hip.BUF1, 0, n=d)
                m.navi_encoder.emit(ch, None, None, navi_pe, dest_feature=dest_f)
                m.add_navi.emit(ch, navi_valid, mask_is_valid=True)
                mid = torch.empty_like(x)
                ch.store(hip.BUF1, 0, d, mid)
                ch.run(rows)
                ch = hip.Chain(16, 4 * d + 4)
                ch.load(mid, hip.BUF1, 0, n=d)
                m.add_latent.emit(ch, lat_inv, z)
                m.action_head.emit(ch, type_mask, out)
                ch.run(rows)
            else:
                prep = dict(navi_pe=navi_pe, type_mask=type_mask)
                hd = m._heads_tile_raw(prep, dict(dest_feature=dest_f), z, lat_inv, navi_valid, dict(action_mean=out), rows)
                assert hd is not None
                hip.heads_tile(x, hd)
            if p > 0:
                assert eng.DROP_CTX["site"] == 40 + 12
        finally:
            eng.DROP_CTX = None
        torch.cuda.synchronize()
        outs[name] = out
    ref, got = outs["chain"], outs["tile"]
    none = (ty == 3).to(dev)
    assert float(got[none].abs().max()) == 0.0 and float(ref[none].abs().max()) == 0.0
    scale = float(ref.abs().max())
    assert scale > 1e-3 and float((got - ref).abs().max()) <= 2e-4 * scale, (float((got - ref).abs().max()), scale)


@pytest.mark.parametrize("bf16,K0,K1", [(False, 25, 64), (False, 70, 0), (True, 24, 88), (False, 8, 3)])
def test_attention_lds_ring_equals_small_launches(hip, dev, bf16, K0, K1):
    """The LDS-ring form of the wave-per-row attention kernel (large launches: K / V rows of the next passes in flight as LDS-DMA
    gathers into a per-wave ring, target indices held in registers; csrc/attn.hip) against the same rows launched in chunks below
    1024 rows (4 wavefronts per row, plain gathers): the same per-pair arithmetic, merged in a different order - 1e-5 of the largest
    output. One or two segments, segments past 64 targets (second index register), shared tables (batch_div), bf16 tables,
    rows without a valid target (exact zeros + flag)."""
    g = torch.Generator().manual_seed(K0 * 7 + K1)
    n, S, T0, T1, D = 4, 300, 300, 500, 128
    rows = n * S
    qbuf = torch.randn(rows, 640, generator=g).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    fxy, fyw = H.make_freqs_xy(32, 1e3).to(dev), H.make_freqs_rad(64).to(dev)

    def seg(T, K, div):
        kv = torch.randn((n // div) * T, 256, generator=g)
        kv = (kv.to(torch.bfloat16) if bf16 else kv).to(dev)
        idx = torch.randint(0, T, (n, S, K), generator=g).to(torch.int32).to(dev)
        inv = (torch.rand(n, S, K, generator=g) < 0.3).to(torch.uint8)
        inv[1, 5] = 1
        rel = torch.cat([(torch.rand(n, S, K, 2, generator=g) - 0.5) * 100, (torch.rand(n, S, K, 1, generator=g) - 0.5) * 6], -1).to(dev).contiguous()
        return kv, idx, inv.to(dev), rel, T, div

    specs = [seg(T0, K0, 1)] + ([seg(T1, K1, 2)] if K1 else [])

    def run(r0, r1, out, flag):
        nb = (r1 - r0) // S
        segs = [hip.Seg(kv[(r0 // S // div) * T:], 0, D, T, idx[r0 // S:r1 // S].contiguous(), inv[r0 // S:r1 // S].contiguous(), None, div,
                        rel=rel[r0 // S:r1 // S].contiguous()) for kv, idx, inv, rel, T, div in specs]
        hip.knarpe_attn(qbuf[r0:r1], 0, D, bias, nb, S, segs, out[r0:r1], flag[r0:r1], fxy, fyw)

    big, bflag = torch.full((rows, 640), 3.0, device=dev), torch.full((rows,), 9, dtype=torch.uint8, device=dev)
    run(0, rows, big, bflag)  # 1200 rows: the ring kernel
    small, sflag = torch.full((rows, 640), 5.0, device=dev), torch.full((rows,), 9, dtype=torch.uint8, device=dev)
    for r0 in range(0, rows, 2 * S):  # 600 rows per launch: 4 waves per row (batch entries in pairs: batch_div = 2 stays aligned)
        run(r0, r0 + 2 * S, small, sflag)
    torch.cuda.synchronize()
    assert torch.equal(bflag, sflag) and int(bflag[S + 5]) == 1
    assert float(big[S + 5].abs().max()) == 0.0
    scale = float(small.abs().max())
    assert float((big - small).abs().max()) <= 1e-5 * scale, (float((big - small).abs().max()), scale)


@pytest.mark.parametrize("m,k,n,wt,bias", [(1000, 128, 128, False, True), (70001, 128, 640, False, True), (4097, 640, 128, False, False),
                                           (12345, 256, 128, True, False), (333, 128, 512, True, False), (64, 512, 256, False, True)])
def test_tall_linear_equals_the_library_product(hip, dev, m, k, n, wt, bias):
    """tbx_tall_linear (training's forward / input-gradient products over very many rows, split-bf16 matrix path) vs torch's fp32
    product: < 3e-5 of sum_k |x||w| per output (checked against that bound, computed per output); ragged last workgroups, several K
    chunks / N blocks, the transposed-weight form of the input gradient, a row stride wider than k."""
    g = torch.Generator().manual_seed(m + k + n)
    xw = torch.randn(m, k + 8, generator=g).to(dev)
    x = xw[:, :k]  # (leading dimension k + 8)
    w = (torch.randn(k, n, generator=g) if wt else torch.randn(n, k, generator=g)).to(dev)
    b = torch.randn(n, generator=g).to(dev) if bias else None
    y = hip.tall_linear(x, w, b, wt=wt)
    wm = w.t() if wt else w
    ref = torch.nn.functional.linear(x.double(), wm.double(), None if b is None else b.double())
    bound = torch.nn.functional.linear(x.abs().double(), wm.abs().double())
    torch.cuda.synchronize()
    assert y.shape == (m, n)
    ratio = float(((y.double() - ref).abs() / bound).max())
    assert 0.0 < ratio < 3e-5, ratio


@pytest.mark.parametrize("rows,bf16_tables", [(2048, False), (1031, False), (8192, True), (130, True)])
def test_tl_tail_tile_equals_the_row_chain(tb, hip, dev, rows, bf16_tables):
    """tbx_tl_tail_tile (large launches: the lights' tail - the K/V tables of the agents' 4 light cross-attention layers,
    transformer_rpe.py:220-223 + attention_rpe.py:92-98, and the next-state logits, traffic_light.py:249-286 - as one tile launch on the
    split-bf16 matrix path) vs the exact-fp32 row chain it replaces (TrafficBots.tl_policy's `tl_tail`): K/V rows within 2e-4 of their
    largest entry (bf16 tables: + one bf16 rounding of a slightly different fp32 value), logits within 2e-4 of the clamp range, rows
    of invalid lights exactly 0, the ragged last tile covered; the single-product twin (Schedule.linear_bf16) within 1e-2."""
    eng = import_module("trafficbots_amd.engine")
    m = _default_model(tb, dev).model
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 128, generator=g).to(dev)
    inv = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    d = 128
    layers = m.ag_encoder.tl_kv_layers()
    lins = [t[0] for t in m.tl_state_predictor.mlp.linear_layers()]
    pw = lambda w, b: hip.packed_weight(w, b, mfma32=True)
    w3, b3 = hip.stacked_linear([lins[2]], pad_out_to=16)
    outs = {}
    for name in ("chain", "tile", "tile_bf16"):
        kv = torch.zeros(rows, 2 * d * len(layers), dtype=torch.bfloat16 if bf16_tables else torch.float32, device=dev)
        logits = torch.full((rows, 5), 7.0, device=dev)
        if name == "chain":
            ch = hip.Chain(16, 4 * d + 4)
            ch.load(x, hip.BUF1, 0, n=d)
            eng.emit_kv_tables(ch, layers, kv)
            m.tl_state_predictor.emit(ch, inv, logits)
            ch.run(rows)
        else:
            lights = dict(kv_images=[pw(at.in_proj_weight[d:], at.in_proj_bias[d:]) for _, at in layers],
                          norms=[(nm.weight, nm.bias, nm.eps) for nm, _ in layers], kv_out=kv,
                          mlp_images=[pw(lins[0].weight, lins[0].bias), pw(lins[1].weight, lins[1].bias), pw(w3, b3)], tl_invalid=inv,
                          logits_out=logits, clamp=(-3.0, 3.0))
            with eng.use(eng.DEFAULT.replace(linear_bf16=(name == "tile_bf16"))):
                hip.tl_tail_tile(x, lights)
        torch.cuda.synchronize()
        outs[name] = (kv.float(), logits)
    (kv_c, lg_c), (kv_t, lg_t), (kv_1, lg_1) = outs["chain"], outs["tile"], outs["tile_bf16"]
    scale = float(kv_c.abs().max())
    assert scale > 0.1 and float(lg_c.abs().max()) <= 3.0 and float(lg_c[inv.bool()].abs().max()) == 0.0
    assert float(lg_t[inv.bool()].abs().max()) == 0.0 and float(lg_1[inv.bool()].abs().max()) == 0.0
    tol_kv = (8e-3 if bf16_tables else 2e-4) * scale  # (bf16 tables: one ulp of bf16 = 2^-8 relative where the fp32 values straddle a rounding boundary)
    assert float((kv_t - kv_c).abs().max()) <= tol_kv, (float((kv_t - kv_c).abs().max()), scale)
    assert float((lg_t - lg_c).abs().max()) <= 2e-4 * 3.0
    assert float((kv_1 - kv_c).abs().max()) <= 1.2e-2 * scale and float((lg_1 - lg_c).abs().max()) <= 1.2e-2 * 3.0
    assert float((kv_1 - kv_t).abs().max()) > 0.0  # the single-product twin really is other arithmetic
