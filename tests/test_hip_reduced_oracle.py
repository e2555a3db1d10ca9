"""The bf16-ARITHMETIC schedule (engine.Schedule.reduced(): BASELINE configs[1] says bf16, the reference trains and validates at
precision 16, configs/trainer/default.yaml:16) pinned to the ORACLE op by op, at the sizes its numbers are quoted on.

What runs under reduced():
  * <= 192 agent rows / <= 384 light rows in a launch: `dec_layer_mf1_kernel` (tbx_knarpe_dec_layer with tail_mfma32 = 2: every LINEAR of a
    decoder layer as ONE bf16 product, bf16 K/V tables gathered by the VALU sweeps);
  * above: `tbx_layer_tile_bf16` (row-local chains, one bf16 product per LINEAR) + `tbx_knarpe_attn_fwd_mfma` (bf16 operands on the matrix cores);
  * at any size `tbx_window_tile_bf16` (window PointNets) and, for large launches, `tbx_heads_tile_bf16`.
Each is compared here with `oracle/hptr_ops.py` (= modules/transformer_rpe.py:207-245, attention_rpe.py:137-190, mlp.py:20-72,
polyline_encoder.py:49-61 of the reference, fp32 CPU) on identical inputs. The fp32-class default schedule runs beside it as the control
(same harness, tolerance of the fp32 suite), so a difference is the arithmetic's, not the harness's.

Tolerances: every bound below is <= 2 x the largest difference MEASURED on MI355X (each test prints what it measured;
profiles/r05_reduced_tolerances.txt keeps the log), expressed relative to the largest reference entry of the compared tensor. Where
they come from: bf16 operands carry 2^-9 relative rounding each; a LINEAR of k = 128..512 terms accumulates in fp32, so its output
error is ~2^-9 * sqrt(k) * |x||w| typical; LayerNorm rescales rows to O(1) between stages, so errors add per stage rather than
compound: ~1e-2 of the row scale after a 3-layer block. Integer results (K-nearest sets, masks, light states, flags) must be identical.
Run on the MI355X box with `pytest -m gpu`."""
from importlib import import_module

import pytest
import torch

from oracle import hptr_ops as H
from oracle import trafficbots_oracle as O
from test_hip_rollout import _oracle_tokens, _setup

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# measured on MI355X (round 5): see the docstring. key = (kernel path, shape tag) -> bound relative to max |ref|
# measured (max |d| / max |ref|): default 7.2e-6 .. 7.7e-6 at all four shapes; reduced 3.9e-3 (agents, 64 rows), 4.4e-3 (lights, 128 rows),
# 4.0e-3 (4 scenes, 256 rows: tile kernels + matrix-core attention), 4.6e-3 (WOSAC shape, 4096 rows)
BLOCK_TOL = {
    ("default", "agents_c2"): 1.5e-5, ("default", "lights_c2"): 1.5e-5, ("default", "agents_4sc"): 1.5e-5, ("default", "wosac"): 1.5e-5,
    ("reduced", "agents_c2"): 8e-3, ("reduced", "lights_c2"): 9e-3, ("reduced", "agents_4sc"): 8e-3, ("reduced", "wosac"): 9e-3,
}


def _block(tb, dev, n_layer, seed):
    M = import_module("trafficbots_amd.models.modules.transformer_rpe")
    blk = M.TransformerBlockRPE(n_layer=n_layer, mode="dec_cross_attn", d_rpe=128, d_model=128, n_head=4, k_feedforward=4, dropout_p=0.1,
                                bias=True, activation="relu", out_layernorm=False, apply_q_rpe=False)
    tb.utils.det_fill(blk, seed)
    P = {"t." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    return blk.to(dev).eval(), P


# (tag, n, S, Ks, T, K, live_limit): the agents' block of configs[1] (64 rows: one-launch layer), the lights' block of configs[1] (128
# rows: one-launch layer), 4 scenes of 64 agents (256 rows past the agents' 192-row limit: tile kernels + matrix-core attention), and
# the WOSAC shape's 32 x 128 = 4096 rows (K/V tables shared by the 32 rollouts through batch_div)
SHAPES = [("agents_c2", 1, 64, 25, 1024, 64, 192), ("lights_c2", 1, 128, 24, 1024, 24, None), ("agents_4sc", 4, 64, 25, 1024, 64, 192),
          ("wosac", 32, 128, 25, 1024, 64, 192)]


@pytest.mark.parametrize("tag,n,S,Ks,T,K,limit", SHAPES)
def test_reduced_decoder_block_vs_oracle(tb, tag, n, S, Ks, T, K, limit):
    """TransformerBlockRPE (dec_cross_attn, 3 layers) through engine.run_block under Schedule.reduced() and under the default schedule
    against oracle.hptr_ops.transformer_block. The cross targets are `T` tokens per scene whose K/V tables the engine projects before
    the gather (bf16 under reduced()); the oracle gets the gathered tokens and the pair embeddings of the same relative poses."""
    dev = torch.device(DEV)
    eng = import_module("trafficbots_amd.engine")
    hip = import_module("trafficbots_amd.hip")
    PE = import_module("trafficbots_amd.utils.pose_emb")
    n_layer = 3
    blk, P = _block(tb, dev, n_layer, 23)
    g = torch.Generator().manual_seed(S * 7 + n)
    n_tab = 1 if tag == "wosac" else n  # WOSAC shape: the 32 rollouts of ONE scene share its map tokens
    div = n // n_tab
    x0 = torch.randn(n, S, 128, generator=g)
    src_inv = torch.rand(n, S, generator=g) < 0.15
    x0[src_inv] = 0.0
    tokens = torch.randn(n_tab, T, 128, generator=g)

    def knn(n_, T_, K_):
        rel = torch.cat([(torch.rand(n, S, K_, 2, generator=g) - 0.5) * 120, (torch.rand(n, S, K_, 1, generator=g) - 0.5) * 6.2], -1)
        idx = torch.randint(0, T_, (n, S, K_), generator=g)
        m = torch.rand(n, S, K_, generator=g) < 0.3
        m[src_inv] = True
        return idx, m, rel

    i0, m0, r0 = knn(n, S, Ks)
    ic, mc, rc = knn(n, T, K)
    m0[0, 3] = True  # a valid source without a valid self target (its first update is skipped) ...
    mc[0, 5] = True  # ... and one without a valid cross target
    pe = PE.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3)
    fxy, fyw = pe.pe_xy.freqs.clone(), pe.pe_yaw.freqs.clone()
    pe = pe.to(dev)
    # ---- oracle: gathered cross targets [n, S, K, 128] of each entry's scene, pair embeddings from the same relative poses
    tok_n = tokens.repeat_interleave(div, 0)
    with torch.no_grad():
        tgt = H.gather_tokens(tok_n, ic)
        ref = H.transformer_block(P, "t", "dec_cross_attn", n_layer, 4, x0, src_inv, tgt, mc, H.pe_xy_yaw(rc[..., :2], rc[..., 2], fxy, fyw),
                                  i0, m0, H.pe_xy_yaw(r0[..., :2], r0[..., 2], fxy, fyw))
    scale = float(ref.abs().max())
    to = lambda t: t.to(dev).contiguous()
    u8 = lambda t: t.to(torch.uint8).to(dev).contiguous()
    launched = {}
    for name, sched in (("default", eng.DEFAULT), ("reduced", eng.DEFAULT.reduced())):
        calls = {"mfma": 0, "tile": 0, "mid": 0}
        orig = (hip.knarpe_attn_mfma, hip.layer_tile, hip.knarpe_dec_mid)

        def count(key, fn):
            def f(*a, **kw):
                calls[key] += 1
                return fn(*a, **kw)
            return f

        hip.knarpe_attn_mfma, hip.layer_tile, hip.knarpe_dec_mid = count("mfma", orig[0]), count("tile", orig[1]), count("mid", orig[2])
        try:
            with eng.use(sched), eng.live_limit(limit):
                kv = eng.kv_tables(to(tokens.reshape(n_tab * T, 128)), [(l.norm_tgt, l.attn) for l in blk.layers])
                assert kv.dtype == (torch.bfloat16 if name == "reduced" else torch.float32)
                x = to(x0.reshape(n * S, 128)).clone()
                eng.run_block(blk, x, u8(src_inv.reshape(-1)), n, S, eng.SelfKnn(to(i0.int()), u8(m0), rel=to(r0)),
                              cross=lambda l: [hip.Seg(kv, l * 256, l * 256 + 128, T, to(ic.int()), u8(mc), None, div, rel=to(rc))], pose_rpe=pe)
            torch.cuda.synchronize()
        finally:
            hip.knarpe_attn_mfma, hip.layer_tile, hip.knarpe_dec_mid = orig
        launched[name] = dict(calls)
        y = x.view(n, S, 128).cpu()
        assert torch.isfinite(y).all() and float(y[src_inv].abs().max()) == 0.0
        err = float((y - ref).abs().max())
        rms = float((y - ref).pow(2).mean().sqrt())
        print(f"[decoder block vs oracle] {tag} ({n * S} rows) {name}: max |d| {err:.3g} = {err / scale:.3g} of max |ref| {scale:.3g}; rms {rms:.3g}; launches {calls}")
        assert err <= BLOCK_TOL[name, tag] * scale, (name, tag, err, scale)
        if name == "reduced":
            assert err > 1e-5 * scale  # the bf16 arithmetic did run
    # the kernel paths the docstring names are the ones that ran
    small = n * S <= (limit or eng.DEFAULT.live_max)
    assert (launched["reduced"]["mid"] > 0) == small and (launched["reduced"]["mfma"] > 0) == (not small) and (launched["default"]["mfma"] == 0)
    assert (launched["reduced"]["tile"] > 1) == (not small)  # (the first projection is a tile launch at any size)


WINDOW_TOL = {"default": 2.5e-5, "reduced": 1e-2}  # measured 8.1e-6 .. 1.25e-5 / 4.1e-3 .. 5.4e-3 of max |ref|


@pytest.mark.parametrize("G,W,add", [(64, 11, False), (1031, 11, False), (128, 11, True), (300, 7, True)])
def test_reduced_window_pointnet_vs_oracle(tb, G, W, add):
    """tbx_window_tile / tbx_window_tile_bf16 (input encoder + 3 PointNet layers + masked pool of the agents' / lights' history windows)
    against oracle input_encoder + pointnet (modules/input_encoder.py:41-61, polyline_encoder.py:49-61, pooling.py:18-19,38)."""
    dev = torch.device(DEV)
    eng = import_module("trafficbots_amd.engine")
    hip = import_module("trafficbots_amd.hip")
    W_ = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W_.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 3)
    P = {k: v.detach().clone() for k, v in wm.model.state_dict().items()}
    m = wm.to(dev).eval().model
    enc, prefix = (m.tl_encoder, "tl_encoder") if add else (m.ag_encoder, "ag_encoder")
    g = torch.Generator().manual_seed(G + W)
    n_attr = 16 if add else 20
    attr = torch.randn(G, W, n_attr, generator=g)
    pe = torch.randn(G, 128, generator=g) if add else torch.randn(G, W, 64, generator=g)
    inv = torch.rand(G, W, generator=g) < 0.3
    inv[5] = True   # a window without a valid row -> exactly 0
    inv[7, 1:] = True
    with torch.no_grad():
        xin = H.input_encoder(P, prefix + ".input_encoder", "add" if add else "cat", attr[None], pe[None, :, None, :].expand(-1, -1, W, -1) if add else pe[None])
        ref = H.pointnet(P, prefix + ".temp_encoder", xin.masked_fill(inv[None, ..., None], 0.0), inv[None], 3)[0]
    scale = float(ref.abs().max())
    inv8 = inv.reshape(-1).to(torch.uint8).to(dev)
    for name, sched in (("default", eng.DEFAULT), ("reduced", eng.DEFAULT.reduced())):
        out = torch.full((G, 128), 7.0, device=dev)
        with eng.use(sched):
            assert hip.tile_products() == (1 if name == "reduced" else 3)
            if add:
                imgs = enc._window_tile_images()
                hip.window_tile(attr.reshape(G * W, n_attr).to(dev).contiguous(), pe.to(dev), inv8, imgs[0], imgs[1], W, out, add_mode=True)
            else:
                a32 = torch.zeros(G * W, 32)
                a32[:, :20] = attr.reshape(G * W, 20)
                imgs = enc._window_tile_images(32)
                hip.window_tile(a32.to(dev), pe.reshape(G * W, 64).to(dev).contiguous(), inv8, imgs[0], imgs[1], W, out)
        torch.cuda.synchronize()
        assert float(out[5].abs().max()) == 0.0
        err = float((out.cpu() - ref).abs().max())
        print(f"[window PointNet vs oracle] G={G} W={W} add={add} {name}: max |d| {err:.3g} = {err / scale:.3g} of max |ref| {scale:.3g}")
        assert err <= WINDOW_TOL[name] * scale, (name, err, scale)
        if name == "reduced":
            assert err > 1e-6 * scale


HEADS_TOL = {"default": 2e-5, "reduced": 9.5e-3}  # measured 9.2e-6 .. 1.06e-5 / 4.7e-3 .. 4.8e-3 of max |ref|


@pytest.mark.parametrize("rows", [1030, 4096])
def test_reduced_heads_vs_oracle(tb, rows):
    """tbx_heads_tile / tbx_heads_tile_bf16 (navigation + latent adders, the action head's three type branches and their masked sum;
    traffic_bots.py:206-221, add_navi_latent.py:52-65, action_head.py:74-100) against the oracle's add_navi_latent / action_head
    on the same rows (the embedded navigation / latent features are given: the engine hoists `mlp_in(z)` out of the step)."""
    dev = torch.device(DEV)
    eng = import_module("trafficbots_amd.engine")
    hip = import_module("trafficbots_amd.hip")
    W_ = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W_.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 3)
    P = {k: v.detach().clone() for k, v in wm.model.state_dict().items()}
    m = wm.to(dev).eval().model
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 128, generator=g)
    navi_emb, lat_emb = torch.relu(torch.randn(rows, 128, generator=g)), torch.relu(torch.randn(rows, 128, generator=g))
    navi_valid = torch.rand(rows, generator=g) < 0.8
    lat_valid = torch.rand(rows, generator=g) < 0.8
    ty = torch.randint(0, 4, (rows,), generator=g)  # 3: no type (an invalid agent)
    with torch.no_grad():  # oracle: add_navi_latent after its mlp_in (oracle/trafficbots_oracle.py add_navi_latent), then action_head
        def adder(prefix, h, z, zv):
            zi = ~zv
            c = torch.cat([h, z.masked_fill(zi.unsqueeze(-1), 0)], -1)
            return H.mlp(P, prefix + ".mlp", c, end_act=True, mask_invalid=zi) + h

        h = adder("add_latent", adder("add_navi", x[None], navi_emb[None], navi_valid[None]), lat_emb[None], lat_valid[None])
        om = O.TrafficBotsOracle(P, tb.config.default_model_cfg(), training=False)
        ag_type = torch.nn.functional.one_hot(ty.clamp(max=2), 3).bool() & (ty != 3)[:, None]
        ref, _ = om.action_head(h, (ty != 3)[None], ag_type[None])
        ref = ref[0]
    scale = float(ref.abs().max())
    type_mask = torch.stack([(ty != i) for i in range(3)]).to(torch.uint8).contiguous().to(dev)
    navi_d, lat_d = navi_emb.masked_fill(~navi_valid[:, None], 0.0).to(dev), lat_emb.masked_fill(~lat_valid[:, None], 0.0).to(dev)
    pw = lambda w, b, **kw: hip.packed_weight(w, b, mfma32=True, **kw)
    lins = [[t[0] for t in mlp.linear_layers()] for mlp in m.action_head.mlp_mean]
    w1, b1 = hip.stacked_linear([l[0] for l in lins])
    w2, b2 = hip.stacked_linear([l[1] for l in lins])
    w3, b3 = hip.stacked_linear([l[2] for l in lins], pad_out_to=16)
    imgs = [pw(t[0].weight, t[0].bias) for t in m.add_navi.mlp.linear_layers()] + [pw(t[0].weight, t[0].bias) for t in m.add_latent.mlp.linear_layers()]
    imgs += [pw(w1, b1), pw(w2, b2, groups=3), pw(w3, b3, groups=3)]
    for name, sched in (("default", eng.DEFAULT), ("reduced", eng.DEFAULT.reduced())):
        out = torch.full((rows, 2), 7.0, device=dev)
        with eng.use(sched):
            hip.heads_tile(x.to(dev), dict(images=imgs, navi_emb=navi_d, latent_emb=lat_d, navi_valid=navi_valid.to(torch.uint8).to(dev),
                                           latent_invalid=(~lat_valid).to(torch.uint8).to(dev), type_mask=type_mask, action_out=out))
        torch.cuda.synchronize()
        assert float(out[(ty == 3).to(dev)].abs().max()) == 0.0
        err = float((out.cpu() - ref).abs().max())
        print(f"[heads vs oracle] rows={rows} {name}: max |d| {err:.3g} = {err / scale:.3g} of max |ref| {scale:.3g}")
        assert err <= HEADS_TOL[name] * scale, (name, err, scale)
        if name == "reduced":
            assert err > 1e-6 * scale


def test_reduced_attention_module_shapes_vs_oracle(tb):
    """ONE attention call in the matrix-core form (tbx_knarpe_attn_fwd_mfma, bf16 tables) against oracle.hptr_ops.attention_rpe
    (attention_rpe.py:83-198) - projections by the fp32 row chain on both sides of the kernel, so the difference is the attention
    kernel's bf16 operands alone; the partial last quad of the persistent kernel (n * S % 4 != 0) with batch_div > 1 is covered."""
    dev = torch.device(DEV)
    eng = import_module("trafficbots_amd.engine")
    hip = import_module("trafficbots_amd.hip")
    M = import_module("trafficbots_amd.models.modules.attention_rpe")
    PE = import_module("trafficbots_amd.utils.pose_emb")
    att = M.AttentionRPE(d_model=128, n_head=4, dropout_p=0.1, d_rpe=128)
    tb.utils.det_fill(att, 15)
    P = {"a." + k: v.detach().clone() for k, v in att.state_dict().items()}
    att = att.to(dev).eval()
    pe = PE.PoseEmb("pe_xy_yaw", pe_dim=128, theta_xy=1e3)
    fxy, fyw = pe.pe_xy.freqs.clone(), pe.pe_yaw.freqs.clone()
    for n, S, T, K, div in ((6, 67, 300, 40, 3), (32, 128, 1024, 64, 32), (3, 65, 128, 24, 1)):
        g = torch.Generator().manual_seed(n * S)
        src = torch.randn(n, S, 128, generator=g)
        tokens = torch.randn(n // div, T, 128, generator=g)
        idx = torch.randint(0, T, (n, S, K), generator=g)
        mask = torch.rand(n, S, K, generator=g) < 0.3
        mask[0, 2] = True
        rel = torch.cat([(torch.rand(n, S, K, 2, generator=g) - 0.5) * 120, (torch.rand(n, S, K, 1, generator=g) - 0.5) * 6.2], -1)
        with torch.no_grad():
            ref = H.attention_rpe(P, "a", 4, src, H.gather_tokens(tokens.repeat_interleave(div, 0), idx), mask, H.pe_xy_yaw(rel[..., :2], rel[..., 2], fxy, fyw))
        scale = float(ref.abs().max())
        rows = n * S
        to = lambda t: t.to(dev).contiguous()
        for name, sched in (("default", eng.DEFAULT), ("reduced", eng.DEFAULT.reduced())):
            with eng.use(sched):
                # K/V table of the tokens (no LayerNorm: the oracle call above takes the tokens as they are) and [q | W_k^T q] of the sources
                kv = torch.empty(tokens.shape[0] * T, 256, dtype=eng.kv_dtype(), device=dev)
                ch = eng.row_chain(kv.shape[0], 132, 132, 132)
                ch.load(to(tokens.reshape(-1, 128)), hip.BUF1, 0, n=128)
                ch.linear(hip.BUF1, 0, hip.GLOBAL, 0, att.in_proj_weight[128:], att.in_proj_bias[128:], out=kv)
                ch.run(kv.shape[0])
                q = torch.empty(rows, eng.Q_LD, device=dev)
                ch = hip.Chain(16, 772)
                ch.load(to(src.reshape(rows, 128)), hip.BUF0, 0, n=128)
                w = eng.emit_qkv(ch, att, hip.BUF0, 0, hip.BUF0, 128, with_kv=False)
                ch.store(hip.BUF0, 128, w, q)
                ch.run(rows)
                obuf = torch.empty(rows, eng.O_LD, device=dev)
                flag = torch.empty(rows, dtype=torch.uint8, device=dev)
                seg = hip.Seg(kv, 0, 128, T, to(idx.int()), to(mask.to(torch.uint8)), None, div, rel=to(rel))
                eng.attention(q, 0, 128, att, n, S, [seg], obuf, flag, fxy.to(dev), fyw.to(dev))
                out = torch.empty(rows, 128, device=dev)
                ch = hip.Chain(16, 644)
                ch.zero(hip.BUF1, 0, 128)
                eng.emit_attn_out(ch, att, obuf, flag)
                ch.store(hip.BUF1, 0, 128, out)
                ch.run(rows)
            torch.cuda.synchronize()
            y = out.view(n, S, 128).cpu()
            assert float(y[0, 2].abs().max()) == 0.0
            err = float((y - ref).abs().max())
            print(f"[attention vs oracle] n={n} S={S} K={K} div={div} {name}: max |d| {err:.3g} = {err / scale:.3g} of max |ref| {scale:.3g}")
            assert err <= (1.5e-6 if name == "default" else 7e-3) * scale, (name, err, scale)  # measured 5.6e-7 .. 7.6e-7 / 3.1e-3 .. 3.5e-3 of max |ref|


# ---- closed loop over the whole horizon at configs[1]'s size: rollout-level figures against the ORACLE's trajectory for the three
# arithmetic classes. Bounds = 2 x measured on MI355X (printed; profiles/r05_reduced_tolerances.txt).
# measured: exact ADE 0.23 mm / FDE 1.1 mm, default 0.36 mm / 1.4 mm, reduced 0.21 m / 0.69 m (largest displacement over the first 16 / 30 /
# 60 steps 22 mm / 0.17 m / 1.0 m: 2^-9 operand rounding into a loop that amplifies ~10x per 15 free steps); flags and light states 100 % in all
C2_ACCEPT = {  # schedule -> (ADE m, FDE m, flag agreement)
    "exact": (4.5e-4, 2.3e-3, 0.999),
    "default": (7e-4, 2.8e-3, 0.999),
    "reduced": (0.42, 1.4, 0.999),
}


def test_c2_free_rollout_90_steps_contractive_weights_acceptance(tb):
    """configs[1]'s scene (64 agents / 1024 polylines / 128 lights), 10 warm-start + 80 FREE closed-loop steps with every residual branch
    and the action head's output layer scaled by 0.3 (test_hip_rollout.py::test_free_rollout_80_steps_full_gain_contractive_weights at the
    8-agent size): average / final displacement error against the oracle's trajectory and agreement of the logged flags, for the
    exact-fp32 schedule, the default (split-bf16 LINEAR stages) and the bf16-arithmetic schedule."""
    dev = torch.device(DEV)
    E = import_module("trafficbots_amd.engine")
    wm, P, b, bd = _setup(tb, dev, (64, 1024, 128), 32)
    with torch.no_grad():
        for k, p in wm.model.state_dict().items():
            if k.endswith(("linear2.weight", "linear2.bias", "out_proj_weight", "out_proj_bias")) or (
                    k.startswith("action_head.mlp_mean") and ".fc_layers.4." in k):
                p.mul_(0.3)
                P[k] = P[k] * 0.3
    cfg, scfg = tb.config.default_model_cfg(n_tgt_knn=32), tb.config.default_sim_cfg()
    om = O.TrafficBotsOracle(P, cfg, training=False)
    with torch.no_grad():
        mp_o = om.mp_encoder(b["sc/mp_valid"], b["sc/mp_attr"], b["sc/mp_pose"], b["ref/mp_type"])
        tl_o = om.tl_pre_compute(b["sc/tl_valid"], b["sc/tl_attr"], b["sc/tl_pose"], **mp_o)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 64, 16, generator=g)
    valid = b["sc/ag_valid"].any(-1)
    bh = dict(b)
    bh["hist/ag_valid"], bh["hist/ag_pose"], bh["hist/ag_motion"] = b["sc/ag_valid"], b["sc/ag_pose"], b["sc/ag_motion"]
    T = 90
    with torch.no_grad():
        ro = O.Sim(om, scfg, False).rollout(bh, mp_o, tl_o, z, valid, b["gt/ag_navi"], valid, scfg.teacher_forcing_joint_future_pred, T,
                                            gt_prefix="hist", tl_gt_key="sc/tl_state")
    assert float(ro["action"][:, :, 12:].abs().max()) > 1.0  # the loop does move
    for name in ("exact", "default", "reduced"):
        wm.schedule = {"exact": E.DEFAULT.replace(dec_tail_mfma=False, tile_small=False, navi_rider=False), "default": E.DEFAULT,
                       "reduced": E.DEFAULT.reduced()}[name]
        wm.engine_cache = 0
        mp, tl = wm.encode_scene(bd)
        ag_tokens = {"ag_type": bd["ref/ag_type"], "ag_size": bd["ref/ag_size"], "ag_attr": bd["sc/ag_attr"], "gt_valid": bd["sc/ag_valid"],
                     "gt_pose": bd["sc/ag_pose"], "gt_motion": bd["sc/ag_motion"], "ag_latent": z.to(dev), "ag_latent_valid": valid.to(dev),
                     "ag_navi": bd["gt/ag_navi"], "ag_navi_valid": valid.to(dev)}
        buf = wm.rollout(ag_tokens, mp, tl, bd["sc/tl_state"], wm.teacher_forcing_joint_future_pred,
                         wm._rule_checker(bd, bd["gt/ag_navi"], tl), T, True)
        buf.flatten_joint_future(1)
        pv = ro["pred_valid"]
        d = (buf.pred_pose[:, 0, :, :, :2].cpu() - ro["pred_pose"][..., :2]).norm(dim=-1)
        both = pv & buf.pred_valid[:, 0].cpu()
        ade, fde = float(d[both].mean()), float(d[..., -1][both[..., -1]].mean())
        agree = lambda a_, b_: float((a_ == b_).float().mean())
        flags = min(agree(buf.pred_valid[:, 0].cpu(), pv), agree(buf.violation["outside_map"][:, 0].cpu(), ro["outside_map"]),
                    agree(buf.violation["dest_reached"][:, 0].cpu(), ro["dest_reached"]))
        tl_agree = agree(buf.vis_dict["tl_state"][:, 0].cpu(), ro["tl_state"])
        first = {t: float(d[..., :t][both[..., :t]].max()) for t in (16, 30, 60)}
        print(f"[C2 rollout acceptance] {name}: ADE {ade:.4g} m, FDE {fde:.4g} m over 90 steps, flag agreement {flags:.4f}, light states {tl_agree:.4f}, "
              f"max displacement over the first 16 / 30 / 60 steps {first[16]:.3g} / {first[30]:.3g} / {first[60]:.3g} m")
        ade_max, fde_max, flags_min = C2_ACCEPT[name]
        assert ade < ade_max and fde < fde_max and flags >= flags_min, (name, ade, fde, flags)
        assert tl_agree >= (1.0 if name != "reduced" else 0.999), (name, tl_agree)
