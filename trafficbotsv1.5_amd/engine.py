"""Host-side scheduling of the HIP kernels for the HPTR transformer stack.

Everything here only *enqueues* kernels through the C ABI (hip.py) on the current HIP stream; there is no host
synchronisation and no data-dependent branching, so a whole simulation step can be captured in a hipGraph.

Formulation (exact restatements of modules/attention_rpe.py, SURVEY.md §0 finding 5):
  * K/V are projected once per *token* (tables) and gathered by KNN index inside the attention kernel, instead of
    per (src, tgt) pair after the gather;
  * `linear_rpe` is folded into the query side (qt_h = W_rpe_k,h^T q_h) and into the weighted sum
    (sum_t a_t (W_rpe_v e_t + b) = W_rpe_v (sum_t a_t e_t) + b  with sum_t a_t = 1 in eval mode).
"""
import contextlib
import dataclasses
import os
import threading
from typing import Callable, List, Optional, Sequence

import torch

from . import hip
from .hip import AUX, BUF0, BUF1, GLOBAL, Chain, Seg

D, NH, DH = 128, 4, 32
QKV_LD, Q_LD, O_LD = 896, 640, 640  # [q|k|v|qt] , [q|qt] , [sum a v | sum a e per head]


def _u8(mask: torch.Tensor) -> torch.Tensor:
    return mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)


def _env(name: str, default: str) -> str:
    return os.environ.get(name, default)


@dataclasses.dataclass
class Schedule:
    """Which launches a piece of the hot path is scheduled as. One object per owner (a `WaymoMotion` module, a `RolloutEngine`, a
    test), made current for the duration of that owner's calls by `use(...)`: nothing here is process-wide state, so engines with
    different schedules (fp32 and bf16 tables, one-stream and two-stream order) live side by side in one process / on 8 ranks. Every
    choice yields the same results up to the tolerances stated in tests/ (most of them bit-identical: see the field comments).
    The defaults come from the TBX_* environment variables, read once at import (`Schedule.from_env`)."""
    # residual updates folded into the producing LINEAR stage (accumulate into the token row, masked rows skipped): 5 stages fewer
    # per decoder layer. False: separate ROWMASK / ADD stages (results differ by fp32 rounding order only).
    fused_residual: bool = True
    load2: bool = True  # token rows + attention output loaded by one stage
    # inference: the attention kernel applies the value half of linear_rpe in its epilogue (tbx_knarpe_attn_fwd_folded): 128 floats
    # per row leave it instead of 640 and the grouped fold stage of the following chain disappears.
    attn_fold: bool = True
    # the wave-per-row form of the kernel (>= 1024 rows) has the folded epilogue too. Measured at the WOSAC shape (4096 rows x 104
    # pairs): 52.2 us per launch instead of 46.3, the following chain 39.8 us instead of 42.4: 1.159 ms per step instead of 1.143.
    attn_fold_big: bool = False
    # a dec_cross_attn layer's [self attention -> out-proj -> LN -> q -> W_k^T q -> cross attention] as ONE launch (tbx_knarpe_dec_mid)
    dec_mid: bool = True
    dec_layer: bool = True   # ... and the whole layer (that launch + the chain after it) as ONE launch, tbx_knarpe_dec_layer
    heads_tail: bool = True  # ... and, for the agents' last layer, the heads (navigation / latent adders + action head) in that launch
    tail_split: bool = True  # a last layer whose chain carries a caller's tail: the layer as one launch + the tail as a short chain
    rowzero: bool = True     # a layer's closing x[invalid] = 0 inside its last LINEAR stage
    masked_groupmax: bool = True
    big_rows: int = 16384    # from here on 32-row tiles + direct-to-global outputs win (measured: +8 % at 16k rows, -10 % at 4k / 64)
    # launches of a few hundred rows in all (one 64-agent scene) are latency-bound: they run as tbx_rowchain_live programs - tiles
    # of live_rows rows, LINEAR stages as v_fma chains per output column (bit-identical to the MFMA tiles) - up to live_max rows.
    live_rows: int = 1
    # Measured on the scenes-per-GPU curve (profiles/r04_scene_curve.json, 64 agents / 128 lights per scene): the one-row-per-
    # workgroup layers win up to 384 rows for the lights' block (48 pairs per row; re-measured at the end of round 4 with the faster
    # layer: 3 scenes = 384 light rows 675 -> 742 k agent-steps/s; 512 rows: the tile kernels) and up to 192 rows for the agents' block
    # (114 pairs per row); from there the tile kernels + the wave-per-row attention do (4 scenes: 0.398 -> 0.309 ms per step, 8 scenes:
    # 0.470 -> 0.332). live_max_agents applies inside TrafficBots.agent_policy (engine.live_limit).
    live_max: int = 384
    live_max_agents: int = 192
    # BASELINE config 2 names bf16: the K/V tables every attention call gathers from stored as bfloat16 (529 B per pair instead of
    # 1041); queries, pose embeddings, scores, softmax and all sums stay fp32. Off: the fp32 parity path.
    kv_bf16: bool = False
    # small launches: the transformer's first projection inside the launch that pools its input rows (TBX_F_POOL_KEEP)
    pool_proj: bool = False
    # LINEAR stages of MFMA row chains (launches past live_max rows) on the split-bf16 matrix path: activations and weights as bf16
    # hi + lo, three products per k on the bf16 MFMA (< 3e-5 of sum |x||w|; tests/test_hip_parity.py)
    split_bf16: bool = False
    # launches past live_max rows (no keyed dropout): a layer's row-local chains as tbx_layer_tile launches - straight-line 16-row
    # tiles, LINEAR on the split-bf16 matrix path (< 3e-5 of sum |x||w| per output) - instead of tbx_rowchain programs (exact fp32)
    tile_layer: bool = True
    tile_min_rows: int = 193
    # ... and, at ANY size, the temporal PointNets of the agents' / lights' windows (tbx_window_tile) and a block's first projection
    # (tbx_layer_tile): at a few hundred rows these are 6-9 dependent stages whose latency the tile kernels cut 3-4x (inference only)
    tile_small: bool = True
    # launches on the tile path: the lights' tail (K/V tables of the agents' 4 layers + next-state logits) as one tbx_tl_tail_tile launch
    # instead of a 16-row exact-fp32 row chain (48 us per step at 2,048 light rows, 97 us at 8,192: profiles/r05_s16_kernel_stats.md)
    tl_tail_tile: bool = True
    prime_graph: bool = True  # RolloutEngine.restore() / refill(): the lights' first pass + first tbx_agent_prep replayed as a graph
    knn_aux_big: bool = True  # large launches: the K-nearest searches on the auxiliary stream beside the window PointNet
    front_big: bool = False  # ... at large launches too - measured SLOWER at the WOSAC shape (6.30 -> 4.56 M agent-steps/s: two pooled rows per
    # workgroup pull the projection's four 64 KiB weight units through every workgroup's L2 port), kept as a switch only
    front_fused: bool = True  # small launches: window PointNet + first projection (+ rider) + the K-nearest searches as ONE launch (tbx_front)
    fused_tail: bool = True  # the agents' tbx_sim_step and the next step's tbx_agent_prep in the tail of the last decoder layer's launch (with the heads)
    navi_rider: bool = True  # small launches: the heads' navigation embedding in extra workgroups of the agents' first-projection launch (no auxiliary stream in the step)
    knn_main: bool = True  # the agents' K-nearest searches on the stepping stream (no cross-queue wait in front of the first attention launch); the auxiliary stream keeps the navigation embedding, joined before the LAST layer
    sim_before_join: bool = True  # the agents' tbx_sim_step on their own stream before the lights' stream is joined
    dec_tail_mfma: bool = True  # tbx_knarpe_dec_layer's LINEAR stages (folds, projections, FFN, heads) on the split-bf16 matrix path
    pe_rides: bool = True       # tbx_knn_embed_multi_pe: the navigation pose embedding in the searches' launch
    # ---- RolloutEngine
    tl_prep_rides: bool = True  # tbx_tl_prep inside the lights' tbx_sim_step launch
    lights_ahead: bool = True   # False: the sequential order on one stream (tl encoder -> agents -> tbx_sim_step)
    # Small launches (both blocks on the one-launch decoder layer: <= live_max light rows, <= live_max_agents agent rows): the step on ONE
    # queue - the lights' and the agents' tbx_front as one launch (tbx_front_pair), their layer l as one launch (tbx_knarpe_dec_layer_pair),
    # the lights' tbx_sim_step in the tail of their last layer (tbx_tl_tail_t.sim_state) as the agents' is (fused_tail): 5 launches per
    # step and no cross-queue edge (the join of the two-stream form costs ~8 us per step: DESIGN.md 8.1). Same launches' arithmetic:
    # bit-identical rollouts (tests/test_hip_rollout.py).
    one_queue: bool = True
    # steps per multi-step graph (even; 1: off). A replay boundary costs a few us of idle device: 4 -> 197.9 k, 16 -> 199.4 k, 40 ->
    # 200.4 k agent-steps/s at the 64-agent scene. Capturing g steps costs g eager steps of host time, so the default suits an engine
    # that runs ONE 80-step rollout; a caller that replays an engine many times raises it (bench.py: 40).
    graph_steps: int = 4
    # the light recurrence reads no agent and no latent, so the K rollouts of a scene carry K identical copies of it: the engine
    # steps the lights once per scene and the agents of the K rollouts attend to that one copy. Bit-identical rollouts.
    share_lights: bool = True
    hoist_constants: bool = True  # False: the heads chain re-embeds the latent / destination feature every step (same values)
    # launches a wavefront takes a whole source row of (inference): the attention on the bf16 matrix cores
    # (tbx_knarpe_attn_fwd_mfma, csrc/attn_mfma.hip): bf16 operands with fp32 accumulation - part of the bf16-ARITHMETIC schedule
    # (Schedule.reduced(); tests/test_hip_attn_mfma.py). False: the fp32 VALU kernel
    attn_mfma: bool = False
    linear_bf16: bool = False       # ONE bf16 product per LINEAR - weights and activations rounded to bfloat16, half the weight bytes a stage streams
                                    # through its CU's L1 port - in the one-launch decoder layer (tail_mfma32 = 2; with kv_bf16 only) and the tile
                                    # kernels (tbx_layer_tile_bf16 / tbx_heads_tile_bf16 / tbx_window_tile_bf16; inference launches)
    attn_mfma_min_rows: int = 193   # ... from this many source rows (= where the wave-per-row forms start; 4 scenes of 64 agents: 0.318 ->
    # 0.264 ms per step, 8 scenes: 0.336 -> 0.286 against a 1024-row threshold, gpurun_out/r04_mfma_rows_*.txt)

    @classmethod
    def from_env(cls) -> "Schedule":
        """Every switch from its TBX_* environment variable (read once, at import). "1" / "0" for the booleans."""
        on = lambda name: _env(name, "1") != "0"        # default on
        off = lambda name: _env(name, "0") == "1"       # default off
        num = lambda name, d: int(_env(name, str(d)))
        return cls(
            fused_residual=on("TBX_FUSED_RESIDUAL"),
            load2=on("TBX_LOAD2"),
            attn_fold=on("TBX_ATTN_FOLD"),
            attn_fold_big=off("TBX_ATTN_FOLD_BIG"),
            dec_mid=on("TBX_DEC_MID"),
            dec_layer=on("TBX_DEC_LAYER"),
            heads_tail=on("TBX_HEADS_TAIL"),
            tail_split=on("TBX_TAIL_SPLIT"),
            rowzero=on("TBX_ROWZERO"),
            masked_groupmax=on("TBX_MASKED_GROUPMAX"),
            big_rows=num("TBX_BIG_ROWS", 16384),
            live_rows=num("TBX_LIVE_ROWS", 1),
            live_max=num("TBX_LIVE_MAX", 384),
            live_max_agents=num("TBX_LIVE_MAX_AGENTS", 192),
            kv_bf16=off("TBX_KV_BF16"),
            pool_proj=off("TBX_POOL_PROJ"),
            split_bf16=off("TBX_SPLIT_BF16"),
            tile_layer=on("TBX_TILE_LAYER"),
            tile_min_rows=num("TBX_TILE_MIN_ROWS", 193),
            tile_small=on("TBX_TILE_SMALL"),
            tl_tail_tile=on("TBX_TL_TAIL_TILE"),
            prime_graph=on("TBX_PRIME_GRAPH"),
            knn_aux_big=on("TBX_KNN_AUX_BIG"),
            front_big=off("TBX_FRONT_BIG"),
            front_fused=on("TBX_FRONT_FUSED"),
            fused_tail=on("TBX_FUSED_TAIL"),
            navi_rider=on("TBX_NAVI_RIDER"),
            knn_main=on("TBX_KNN_MAIN"),
            sim_before_join=on("TBX_SIM_BEFORE_JOIN"),
            dec_tail_mfma=on("TBX_DEC_TAIL_MFMA"),
            pe_rides=on("TBX_PE_RIDES"),
            tl_prep_rides=on("TBX_TL_PREP_RIDES"),
            lights_ahead=on("TBX_LIGHTS_AHEAD"),
            one_queue=on("TBX_ONE_QUEUE"),
            graph_steps=max(1, num("TBX_GRAPH_STEPS", 4) // 2 * 2),
            share_lights=on("TBX_SHARE_LIGHTS"),
            hoist_constants=os.environ.get("TBX_NO_HOIST") is None,
            attn_mfma=off("TBX_ATTN_MFMA"),
            linear_bf16=off("TBX_LINEAR_BF16"),
            attn_mfma_min_rows=num("TBX_ATTN_MFMA_MIN_ROWS", 193),
        )

    def replace(self, **kw) -> "Schedule":
        return dataclasses.replace(self, **kw)

    def reduced(self) -> "Schedule":
        """The bf16-ARITHMETIC schedule (BASELINE configs[1] says bf16; the reference runs at precision 16,
        configs/trainer/default.yaml:16), as torch's autocast(bfloat16) would run the LINEAR / attention contractions:
          * bfloat16 K/V tables (every attention call; 529 B per pair);
          * launches of >= attn_mfma_min_rows (193) source rows: the attention with bf16 operands on the matrix cores, fp32 accumulation
            and softmax (tbx_knarpe_attn_fwd_mfma); calls outside that kernel's preconditions take the VALU kernel (mfma_attention_ok);
          * EVERY inference LINEAR as ONE bf16 product (weights and activations rounded to bfloat16, fp32 accumulation): the one-launch
            decoder layer (dec_layer_mf1_kernel: launches of <= live_max = 384 rows, <= live_max_agents = 192 for the agents' block) and,
            through hip.tile_products, every tbx_layer_tile / tbx_heads_tile / tbx_window_tile launch (their *_bf16 entry points) - the
            scene encoders (map, lights' pre-compute) included.
        K-nearest searches, dynamics, LayerNorms, softmax and the row chains keep fp32. Tolerances against the ORACLE, op by op and closed
        loop, at the sizes the schedule is quoted on: tests/test_hip_reduced_oracle.py (+ tests/test_hip_attn_mfma.py for the kernel,
        tests/test_hip_rollout.py / test_hip_boundary.py for the teacher-forced and WOSAC-shape loops); table in DESIGN.md 2."""
        return self.replace(kv_bf16=True, attn_mfma=True, linear_bf16=True)


DEFAULT = Schedule.from_env()
_tls = threading.local()


def current() -> Schedule:
    """The schedule of whoever is running (innermost `use`), else the process default."""
    return getattr(_tls, "sched", None) or DEFAULT


hip.tile_products = lambda: 1 if current().linear_bf16 else 3  # (the tile kernels' *_bf16 entry points under Schedule.linear_bf16)


@contextlib.contextmanager
def use(sched: Optional[Schedule]):
    """Make `sched` current for the calls inside the block (None: leave the current one)."""
    prev = getattr(_tls, "sched", None)
    _tls.sched = sched if sched is not None else prev
    try:
        yield current()
    finally:
        _tls.sched = prev


# Keyed dropouts of training for whoever emits chains while it is set (train_graph's stepping pass; None in inference):
# dict(seed=int64[1] device tensor, site=last elementwise site id used, call=last attention call id used, step=closed-loop step).
# The emitters below take their ids from it in execution order - the order train_graph's torch ops take theirs.
DROP_CTX: Optional[dict] = None


def drop_site(p: float):
    """(p, seed, site, step) for a Chain.dropout stage, or None (inference / p = 0)."""
    if DROP_CTX is None or not (p is not None and p > 0):
        return None
    DROP_CTX["site"] += 1
    return (float(p), DROP_CTX["seed"], DROP_CTX["site"], DROP_CTX["step"])


def drop_call(attn):
    """hip.knarpe_attn's `drop` argument for an attention module, or None."""
    if DROP_CTX is None or not attn.dropout_p > 0:
        return None
    DROP_CTX["call"] += 1
    return (float(attn.dropout_p), DROP_CTX["seed"], DROP_CTX["call"], 1, DROP_CTX["step"])


def mfma_attention_ok(qbuf, q_off: int, qt_off: int, segs, obuf) -> bool:
    """tbx_knarpe_attn_fwd_mfma's preconditions (csrc/attn_mfma.hip: TBX_ERR_UNSUPPORTED / TBX_ERR_ALIGN cases), checked on the host so
    that a call outside them takes the VALU kernel instead of raising mid-step: one or two segments of <= 128 targets in all, tables of
    ONE dtype with 8-element-aligned leading dimension and offsets, 16-byte-aligned bases."""
    if not (1 <= len(segs) <= 2) or sum(sg.k for sg in segs) > 128:
        return False
    if qbuf.stride(0) % 4 or q_off % 4 or qt_off % 4 or obuf.stride(0) % 4 or qbuf.data_ptr() % 16 or obuf.data_ptr() % 16:
        return False
    for sg in segs:
        if sg.kv.dtype != segs[0].kv.dtype or sg.kv.stride(0) % 8 or sg.k_off % 8 or sg.v_off % 8 or sg.kv.data_ptr() % 16:
            return False
        if sg.n_tgt * sg.kv.stride(0) * 4 >= 1 << 32:
            return False
    return True


def attention(qbuf, q_off: int, qt_off: int, attn, n: int, S: int, segs, obuf, flag, fxy, fyw, drop=None, fold=None):
    """One KNARPE attention call (attention_rpe.py:137-190): hip.knarpe_attn, or - large inference launches whose segments are all
    given as relative poses, Schedule.attn_mfma - the matrix-core form (same output rows, its own rounding)."""
    if (current().attn_mfma and fold is None and n * S >= current().attn_mfma_min_rows and obuf.shape[1] >= D + NH * D and fxy is not None
            and all(sg.rel is not None and sg.emb is None for sg in segs) and mfma_attention_ok(qbuf, q_off, qt_off, segs, obuf)):
        hip.knarpe_attn_mfma(qbuf, q_off, qt_off, n, S, segs, obuf, flag, fxy, fyw, drop=drop)  # (drop: training's stepping pass, keyed mask)
        return
    hip.knarpe_attn(qbuf, q_off, qt_off, attn.linear_rpe.bias, n, S, segs, obuf, flag, fxy, fyw, drop=drop, fold=fold)


def emit_qkv(ch: Chain, attn, src_buf: int, src_col: int, dst_buf: int, dst_col: int, with_kv: bool) -> int:
    """[q | k | v | qt] (896 cols) or [q | qt] (640 cols) of the tile at dst_col. attention_rpe.py:92-98,147."""
    w_in, b_in = attn.in_proj_weight, attn.in_proj_bias
    nq = 3 * D if with_kv else D
    ch.linear(src_buf, src_col, dst_buf, dst_col, w_in[:nq], b_in[:nq])
    # qt_h = q_h @ W_rpe_k[h*32:(h+1)*32, :]  for the 4 heads as one block-diagonal stage
    ch.linear(dst_buf, dst_col, dst_buf, dst_col + nq, attn.linear_rpe.weight[:D], wt=True, groups=NH, src_stride=DH, dst_stride=D)
    return nq + NH * D






def attn_fold_image(attn) -> torch.Tensor:
    return hip.packed_weight(attn.linear_rpe.weight[D:], attn.linear_rpe.bias[D:], groups=NH, gemv=True)


def emit_attn_out(ch: Chain, attn, obuf: torch.Tensor, row_no_valid: torch.Tensor, x_buf: int = BUF1, drop=None, x: Optional[torch.Tensor] = None):
    """x += out_proj(sum a v + W_rpe_v (sum a e) + b_rpe_v), zero for rows without a valid target.
    attention_rpe.py:152,182-190; transformer_rpe.py:212-213,233. drop = (p, seed, site, step): the residual dropout of training
    (transformer_rpe.py:56-60) as a keyed DROPOUT stage. x: the token rows [rows, 128] are not in x_buf yet - they are loaded in the
    same stage as the attention output (TBX_F_LOAD2: one memory round trip for both)."""
    folded = obuf.shape[1] == D  # the attention kernel already applied the fold
    if x is not None and current().load2:
        ch.load2(obuf, BUF0, 0, x, x_buf, 0)
    else:
        if x is not None:
            ch.load(x, x_buf, 0, n=D)
        ch.load(obuf, BUF0, 0, n=obuf.shape[1])
    if not folded:
        # per head: (sum a v)_h += W_rpe_v,h (sum a e)_h + b_rpe_v,h, one block-diagonal stage
        ch.linear(BUF0, D, BUF0, 0, attn.linear_rpe.weight[D:], attn.linear_rpe.bias[D:], accum=True, groups=NH, src_stride=D,
                  dst_stride=DH)
    if drop is None and ch.pack_weights and current().fused_residual:
        # one stage: x += rows without a valid target ? 0 : out_proj(...)  (TBX_F_ROWSKIP + accumulate into the token row)
        ch.linear(BUF0, 0, x_buf, 0, attn.out_proj_weight, attn.out_proj_bias, accum=True, skip_rows=row_no_valid)
        return
    ch.linear(BUF0, 0, AUX, 0, attn.out_proj_weight, attn.out_proj_bias)
    ch.rowmask(AUX, 0, D, mask=row_no_valid)
    if drop is not None:
        ch.dropout(AUX, 0, D, *drop)
    ch.add(AUX, 0, x_buf, 0, D)


def emit_ffn(ch: Chain, layer, x_buf: int = BUF1, drop_hidden=None, drop_out=None, zero_rows: Optional[torch.Tensor] = None) -> bool:
    """x += linear2(relu(linear1(norm2(x)))). transformer_rpe.py:234-237; drop_* = (p, seed, site, step) in training.
    zero_rows (u8 per row): the layer's closing `x[invalid] = 0` folded into the last stage where that stage is the fused residual
    (TBX_F_ROWZERO) - returns True if it was."""
    ch.layernorm(x_buf, 0, BUF0, 0, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    ch.linear(BUF0, 0, BUF0, D, layer.linear1.weight, layer.linear1.bias, relu=True)
    if drop_hidden is not None:
        ch.dropout(BUF0, D, layer.linear1.weight.shape[0], *drop_hidden)
    if drop_out is None and ch.pack_weights and current().fused_residual:
        if zero_rows is not None and current().rowzero:
            ch.linear(BUF0, D, x_buf, 0, layer.linear2.weight, layer.linear2.bias, accum=True, skip_rows=zero_rows, zero_skipped=True)
            return True
        ch.linear(BUF0, D, x_buf, 0, layer.linear2.weight, layer.linear2.bias, accum=True)  # x += linear2(...) in one stage
        return False
    ch.linear(BUF0, D, AUX, 0, layer.linear2.weight, layer.linear2.bias)
    if drop_out is not None:
        ch.dropout(AUX, 0, D, *drop_out)
    ch.add(AUX, 0, x_buf, 0, D)
    return False


def emit_mlp(ch: Chain, mlp, src_buf: int, src_col: int, bufs=(BUF0, BUF1), out_col: int = 0, end_buf: Optional[int] = None):
    """A `MLP` container (modules/mlp.py) as Linear[+LN]+ReLU stages, ping-ponging between two buffers.
    Returns the buffer that holds the output at out_col."""
    lins = mlp.linear_layers()
    cur, col = src_buf, src_col
    for j, (lin, ln, act) in enumerate(lins):
        last = j == len(lins) - 1
        dst = bufs[0] if cur != bufs[0] else bufs[1]
        if last and end_buf is not None:
            dst = end_buf
        if ln is None:
            ch.linear(cur, col, dst, out_col, lin.weight, lin.bias, relu=act)
        else:
            ch.linear(cur, col, dst, out_col, lin.weight, lin.bias)
            other = bufs[0] if dst != bufs[0] else bufs[1]
            ch.layernorm(dst, out_col, dst, out_col, ln.weight, ln.bias, ln.eps)
            if act:
                ch.clamp(dst, out_col, lin.weight.shape[0], 0.0, float("inf"))
            del other
        d = drop_site(mlp.dropout_p)
        if d is not None:
            ch.dropout(dst, out_col, lin.weight.shape[0], *d)
        cur, col = dst, out_col
    return cur




def emit_pointnet(ch: Chain, pl_encoder, row_invalid: torch.Tensor, out: torch.Tensor, x_buf: int = BUF1, keep: bool = False) -> Optional[int]:
    """PointNet over the rows of one group (tile): polyline_encoder.py:49-61 + pooling.py:18-19,38.
    Input x at x_buf[:, 0:128]; pooled row -> out[group]. keep: the pooled rows also stay in LDS and the chain goes on with them
    (Chain.poolmax keep=) - returns the buffer they are in ([:, 0:width])."""
    cur = x_buf
    for mlp in pl_encoder.mlp_layers:
        lin = mlp.linear_layers()[0][0]
        nxt = BUF0 if cur != BUF0 else BUF1
        half = lin.weight.shape[0]
        ch.linear(cur, 0, nxt, 0, lin.weight, lin.bias, relu=True)
        d = drop_site(mlp.dropout_p)
        if d is not None:
            ch.dropout(nxt, 0, half, *d)
        if d is None and current().masked_groupmax:  # one stage: the maximum over the group's valid rows, masked rows zeroed in both halves
            ch.groupmax(nxt, 0, nxt, half, half, mask=row_invalid)
        else:
            ch.rowmask(nxt, 0, half, mask=row_invalid, fill=float("-inf"))
            ch.groupmax(nxt, 0, nxt, half, half)
            ch.rowmask(nxt, 0, 2 * half, mask=row_invalid, fill=0.0)
        cur = nxt
    if not keep:
        ch.poolmax(cur, 0, out.shape[1], out, mask=row_invalid)
        return None
    kept = BUF1 if cur != BUF1 else AUX
    ch.poolmax(cur, 0, out.shape[1], out, mask=row_invalid, keep=(kept, 0))
    return kept


# LDS row widths of the transformer-layer chains: BUF0 holds the wide intermediates (attention output 640, FFN hidden
# 128+512), BUF1 the token row x, AUX one 128-wide temporary; wide OUTPUTS (k|v, qt) go straight to global memory.
# 16-row tiles: 58 KB -> two workgroups per CU; 32-row tiles (large grids): 116 KB, half the weight traffic per row.
LAYER_LDW0, LAYER_LDW1, LAYER_AUX = 644, 132, 132








def kv_dtype():
    return torch.bfloat16 if current().kv_bf16 else torch.float32


@contextlib.contextmanager
def live_limit(max_rows: Optional[int]):
    """Inside the block the live-row schedule ends at `max_rows` rows instead of Schedule.live_max (None: unchanged)."""
    prev = getattr(_tls, "live_max", None)
    _tls.live_max = max_rows if max_rows is not None else prev
    try:
        yield
    finally:
        _tls.live_max = prev


def live_rows_for(rows: int) -> int:
    lim = getattr(_tls, "live_max", None)
    lim = current().live_max if lim is None else lim
    return current().live_rows if (current().live_rows and rows <= lim and DROP_CTX is None) else 0


def row_chain(rows: int, ldw: int, ldw1: Optional[int] = None, ld_aux: Optional[int] = None, big: Optional[tuple] = None) -> Chain:
    """A flat (un-grouped) chain for `rows` rows: live-row tiles for small launches, 16-row MFMA tiles otherwise, `big` = the
    (tile_rows, ldw0, ldw1, ld_aux) layout of grids past current().big_rows if the caller has one."""
    live = live_rows_for(rows)
    if live:  # 4-row LDS tiles + 128 KiB of weight slots: narrow side buffers unless the caller asks for more
        return Chain(16, ldw, ldw if ldw1 is None else ldw1, 132 if ld_aux is None else ld_aux, live_rows=live)
    if big is not None and rows >= current().big_rows:
        return Chain(*big)
    return Chain(16, ldw, ldw1, ld_aux)


def layer_chain(rows: int) -> Chain:
    """Small grids: 16-row tiles, everything staged in LDS (2 x 1028-float buffers, 1 workgroup per CU, fewest stages) - as
    live-row tiles up to current().live_max rows. Large grids: 32-row tiles with asymmetric buffers (116 KB) and wide outputs written
    straight to global memory."""
    if live_rows_for(rows):
        return Chain(16, 1028, 132, 132, live_rows=live_rows_for(rows))  # BUF1 holds the token row only, AUX one 128-wide temporary
    return row_chain(rows, 1028, big=(32, LAYER_LDW0, LAYER_LDW1, LAYER_AUX))


def emit_proj(ch: Chain, rows: int, norm, attn, out: torch.Tensor, with_kv: bool, kv16: Optional[torch.Tensor] = None, x_buf: int = BUF1):
    """LN(x in x_buf) -> [q | k | v | qt] / [q | qt] stored to `out` (+ k | v as bfloat16 to kv16 [rows, 256] if given)."""
    if rows >= current().big_rows:
        emit_proj_to(ch, norm, attn, out, with_kv, x_buf=x_buf, kv16=kv16)
    else:
        ch.layernorm(x_buf, 0, BUF0, 0, norm.weight, norm.bias, norm.eps)
        w = emit_qkv(ch, attn, BUF0, 0, BUF0, D, with_kv=with_kv)
        ch.store(BUF0, D, w, out)
        if kv16 is not None:  # the self-attention K/V table as bfloat16 (q and W_k^T q stay fp32 in `out`)
            ch.store(BUF0, 2 * D, 2 * D, kv16)


def emit_proj_to(ch: Chain, norm, attn, out: torch.Tensor, with_kv: bool, x_buf: int = BUF1, kv16: Optional[torch.Tensor] = None):
    """LN(x) -> q [| k | v] | qt written to `out` ([rows, 896] or [rows, 640]); only q is staged in LDS (BUF0[:, 128:256],
    it feeds the per-head rpe fold), k|v and qt go straight to global memory. attention_rpe.py:92-98,147."""
    w_in, b_in = attn.in_proj_weight, attn.in_proj_bias
    ch.layernorm(x_buf, 0, BUF0, 0, norm.weight, norm.bias, norm.eps)
    ch.linear(BUF0, 0, BUF0, D, w_in[:D], b_in[:D])
    ch.store(BUF0, D, D, out, 0)
    nq = D
    if with_kv:
        if kv16 is not None:
            ch.linear(BUF0, 0, GLOBAL, 0, w_in[D:], b_in[D:], out=kv16)  # k | v straight to the bfloat16 table
        else:
            ch.linear(BUF0, 0, GLOBAL, D, w_in[D:], b_in[D:], out=out)
        nq = 3 * D
    ch.linear(BUF0, D, GLOBAL, nq, attn.linear_rpe.weight[:D], wt=True, groups=NH, src_stride=DH, dst_stride=D, out=out)


FIRST_PROJ_LDW = 1028  # BUF0 of such a chain: the LayerNorm row + the 896-wide projection


def first_proj_buffers(rows: int, dev, tile_rows: int = 16) -> Optional[dict]:
    """The q | k | v | qt rows (+ the bfloat16 k | v copy) run_block(first_proj=...) starts from when the producer of its input rows
    ran emit_first_proj in its own chain; None where that form is not used (large launches, training's keyed dropout)."""
    if not (current().pool_proj and live_rows_for(rows)) or tile_rows != 16:  # (two 1028-wide buffers of 16 rows: 148 KB of LDS)
        return None
    return dict(qkv=torch.empty(rows, QKV_LD, dtype=torch.float32, device=dev),
                kv16=torch.empty(rows, 2 * D, dtype=torch.bfloat16, device=dev) if current().kv_bf16 else None)


def emit_first_proj(ch: Chain, block, fp: dict, x_buf: int) -> None:
    """Layer 0's projections of `block` on the rows in x_buf[:, 0:128] (the same stages run_block's first chain runs)."""
    l0 = block.layers[0]
    dec = block.mode == "dec_cross_attn"
    emit_proj(ch, 0, l0.norm_src if dec else l0.norm1, l0.attn_src if dec else l0.attn, fp["qkv"], with_kv=True, kv16=fp["kv16"], x_buf=x_buf)


def tile_rows_ok(rows: int, keyed_dropout: bool = False) -> bool:
    """Launches whose row-local work runs as tile kernels (past the live-row sizes). keyed_dropout: the caller's kernel takes the
    keyed dropouts of training's stepping pass (tbx_layer_tile does; the heads / window kernels are inference only)."""
    c = current()
    if DROP_CTX is not None and not keyed_dropout:
        return False
    return c.tile_layer and not live_rows_for(rows) and rows >= c.tile_min_rows and not c.attn_fold_big


def front_ok(rows: int) -> bool:
    """tbx_front for this launch: the small-launch tile schedule, inference."""
    c = current()
    if not (c.front_fused and tile_small_ok()):
        return False
    if tile_rows_ok(rows, keyed_dropout=True):  # large launches: Schedule.front_big
        return c.front_big and DROP_CTX is None
    return rows < 4096


def front_proj_buffers(rows: int, dev) -> dict:
    """The q | k | v | qt rows (+ bf16 k | v copy) tbx_front's projection part writes: run_block(first_proj=...) starts from them."""
    return dict(qkv=torch.empty(rows, QKV_LD, dtype=torch.float32, device=dev),
                kv16=torch.empty(rows, 2 * D, dtype=torch.bfloat16, device=dev) if current().kv_bf16 else None)


def tile_small_ok() -> bool:
    """Inference launches of any size whose window PointNet / first projection run as tile kernels."""
    c = current()
    return c.tile_layer and c.tile_small and DROP_CTX is None and not c.pool_proj


def _tile_drop(*sites) -> Optional[dict]:
    """hip.layer_tile's `drop` from up to three drop_site() results (attention residual, FFN hidden, FFN output)."""
    live = [d for d in sites if d is not None]
    if not live:
        return None
    p, seed, _, step = live[0]
    return dict(p=p, seed=seed, step=step, sites=tuple(None if d is None else d[2] for d in sites) + (None,) * (3 - len(sites)))


def _img(w, b=None, **kw):
    return hip.packed_weight(w, b, mfma32=True, **kw)


def tile_attn_part(attn, obuf, flag) -> dict:
    return dict(out=obuf, row_no_valid=flag, fold=_img(attn.linear_rpe.weight[D:], attn.linear_rpe.bias[D:], groups=NH),
                out_proj=_img(attn.out_proj_weight, attn.out_proj_bias))


def tile_ffn_part(layer, src_invalid) -> dict:
    return dict(norm2=(layer.norm2.weight, layer.norm2.bias, layer.norm2.eps), linear1=_img(layer.linear1.weight, layer.linear1.bias),
                linear2=_img(layer.linear2.weight, layer.linear2.bias), src_invalid=src_invalid)


def tile_proj_part(norm, attn, out, with_kv: bool, kv16=None) -> dict:
    nq = 3 * D if with_kv else D
    return dict(norm=(norm.weight, norm.bias, norm.eps), image=_img(attn.in_proj_weight[:nq], attn.in_proj_bias[:nq]),
                qfold=_img(attn.linear_rpe.weight[:D], None, wt=True, groups=NH), n=nq, out=out, kv16=kv16)


class SelfKnn:
    """KNN set among the source tokens themselves: idx i32 / invalid u8 [n,S,K] and either the materialised pose embedding
    emb f32 [n,S,K,128] or the relative pose rel f32 [n,S,K,3] (embedding rebuilt inside the attention kernel)."""

    def __init__(self, idx, invalid, emb=None, rel=None):
        self.idx, self.invalid = idx.contiguous(), _u8(invalid).contiguous()
        self.emb = None if emb is None else emb.contiguous()
        self.rel = None if rel is None else rel.contiguous()


def run_block(block, x: torch.Tensor, src_invalid: torch.Tensor, n: int, S: int, self_knn: Optional[SelfKnn],
              cross: Optional[Callable[[int], Sequence[Seg]]] = None, tail: Optional[Callable[[Chain], None]] = None,
              tile_rows: int = 16, pose_rpe=None, drop: Optional[dict] = None, freqs=None, join_stream=None,
              heads_tail: Optional[dict] = None, first_proj: Optional[dict] = None, join_late: bool = False,
              after_first_proj=None, proj_rider=None, tail_mf=None) -> bool:
    """Runs a TransformerBlockRPE (modes enc_self_attn / dec_cross_attn, transformer_rpe.py:48-135,207-245) over the
    token matrix x [n*S, 128] IN PLACE (join_stream: a stream the K-nearest sets are being produced on, waited for right before the
    first attention call). `cross(l)` yields the cross-attention segments of layer l; `tail(chain)`
    appends row-local stages to the last layer's chain (x is in BUF1[:, 0:128] at that point).
    drop (the stepping pass of training, train_graph.py): dict(p=residual / FFN dropout, seed=int64[1] device tensor, site=last
    elementwise site id used, call=last attention call id used, step=closed-loop step): the keyed dropouts of training - in
    the attention kernels and as DROPOUT stages - with the ids train_graph.transformer_block gives them (per layer: call + 1 [,
    call + 2], sites + 1 .. + 3 [+ 4] in execution order)."""
    assert x.shape == (n * S, D) and x.is_contiguous()
    fxy = None if pose_rpe is None else pose_rpe.pe_xy.freqs
    fyw = None if pose_rpe is None else pose_rpe.pe_yaw.freqs
    if freqs is not None:  # (pe_xy.freqs, pe_yaw.freqs) given directly
        fxy, fyw = freqs
    global DROP_CTX
    outer = DROP_CTX
    if drop is not None:  # explicit ids (train_graph.transformer_block)
        DROP_CTX = dict(seed=drop["seed"], site=drop["site"], call=drop["call"], step=drop["step"])
    p_res = drop["p"] if drop is not None else block.dropout_p
    next_site = lambda: drop_site(p_res)
    next_call = drop_call
    rows = n * S
    dev = x.device
    src_invalid = _u8(src_invalid).reshape(-1).contiguous()
    qkv = torch.empty(rows, QKV_LD, dtype=torch.float32, device=dev) if first_proj is None else first_proj["qkv"]
    # (a one-launch layer writes the next layer's q | k | v | qt rows while other workgroups still gather this layer's K / V rows)
    qkv_alt = torch.empty_like(qkv) if current().dec_layer and current().dec_mid and current().attn_fold and bool(live_rows_for(rows)) else None
    kv16 = torch.empty(rows, 2 * D, dtype=torch.bfloat16, device=dev) if current().kv_bf16 and drop is None and DROP_CTX is None else None
    if first_proj is not None:  # layer 0's projections were made by the launch that produced x (emit_first_proj)
        assert drop is None and DROP_CTX is None and (first_proj["kv16"] is not None) == (kv16 is not None)
        kv16 = first_proj["kv16"]
    kv16_alt = torch.empty_like(kv16) if (kv16 is not None and qkv_alt is not None) else None
    fold = current().attn_fold and drop is None and DROP_CTX is None and (bool(live_rows_for(rows)) or current().attn_fold_big)
    obuf = torch.empty(rows, D if fold else O_LD, dtype=torch.float32, device=dev)
    flag = torch.empty(rows, dtype=torch.uint8, device=dev)
    heads_done = False  # -> True if the last layer's launch also ran `heads_tail`
    layers = list(block.layers)
    dec = block.mode == "dec_cross_attn"
    if not dec and block.mode != "enc_self_attn":
        raise NotImplementedError(f"TransformerBlockRPE mode {block.mode} is not on the default hot path")
    q2 = torch.empty(rows, Q_LD, dtype=torch.float32, device=dev) if dec else None

    def first_attn(l):
        return layers[l].attn_src if dec else layers[l].attn

    def first_norm(l):
        return layers[l].norm_src if dec else layers[l].norm1

    tile = tile_rows_ok(rows, keyed_dropout=True)
    if first_proj is None and (tile or tile_small_ok()):
        r = proj_rider() if callable(proj_rider) else proj_rider  # (a side job of the caller's in the same launch: tbx_layer_tile_t.rider_*)
        hip.layer_tile(x, proj=tile_proj_part(first_norm(0), first_attn(0), qkv, True, kv16), store_x=False, rider=r)
    elif first_proj is None:
        assert proj_rider is None
        ch = layer_chain(rows)
        ch.load(x, BUF1, 0, n=D)
        emit_proj(ch, rows, first_norm(0), first_attn(0), qkv, with_kv=True, kv16=kv16)
        ch.run(rows)
    if after_first_proj is not None:
        # the caller forks its auxiliary stream HERE: hipGraph's executor keeps the first-captured child of a node on the node's queue
        # and moves later ones to another (measured, profiles/r03_c2_two_stream_timeline.txt), so this stream's next launch has to
        # be captured before the side work or the whole layer sequence moves queues behind a ~12 us cross-queue wait
        after_first_proj()
    if callable(heads_tail):  # (resolved late: the caller's side work above may be what makes the heads' inputs)
        heads_tail = heads_tail(mfma32=current().dec_tail_mfma and DROP_CTX is None)
    if join_stream is not None and not join_late:  # whoever produced the K-nearest sets on another stream is joined here, not before the projection
        torch.cuda.current_stream().wait_stream(join_stream)
    mid = fold and dec and current().dec_mid and bool(live_rows_for(rows))  # the one-launch attention half: small launches only
    for l, layer in enumerate(layers):
        if join_stream is not None and join_late and l + 1 == len(layers):  # (join_late: that stream only made inputs of the heads)
            torch.cuda.current_stream().wait_stream(join_stream)
        a1 = first_attn(l)
        self_seg = (Seg(qkv, D, 2 * D, S, self_knn.idx, self_knn.invalid, self_knn.emb, rel=self_knn.rel) if kv16 is None else
                    Seg(kv16, 0, D, S, self_knn.idx, self_knn.invalid, self_knn.emb, rel=self_knn.rel))
        whole = mid and current().dec_layer and qkv_alt is not None and (l + 1 < len(layers) or tail is None or current().tail_split) and src_invalid is not None
        if whole:
            # ONE launch for the layer (tbx_knarpe_dec_layer): the attention half below, then out_proj / FFN / x[invalid] = 0 and the
            # next layer's projections - the stages of the chain that followed tbx_knarpe_dec_mid, in the same arithmetic
            a2 = layer.attn
            last = l + 1 == len(layers)
            tmf = current().dec_tail_mfma and DROP_CTX is None
            tkw = dict(mfma32=True) if tmf else dict(gemv=True)
            tl_ = dict(out_proj2=hip.packed_weight(a2.out_proj_weight, a2.out_proj_bias, **tkw),
                       linear1=hip.packed_weight(layer.linear1.weight, layer.linear1.bias, **tkw),
                       linear2=hip.packed_weight(layer.linear2.weight, layer.linear2.bias, **tkw),
                       norm2=(layer.norm2.weight, layer.norm2.bias, layer.norm2.eps), src_invalid=src_invalid,
                       mfma32=(2 if (tmf and current().linear_bf16 and kv16 is not None) else tmf))
            if last and heads_tail is not None and current().heads_tail:
                tl_["heads"] = heads_tail  # the agents' heads in this launch too (tbx_heads_tail_t)
                heads_done = True
            tail_in_launch = False
            if last and tmf and tail_mf is not None and "heads" not in tl_:
                # the caller's tail as tbx_tl_tail_t fields (a callable: built only where it is used; None: not of that shape)
                lights = tail_mf() if callable(tail_mf) else tail_mf
                if lights is not None:
                    tl_["lights"] = lights
                    tail_in_launch = True
            if not last:
                an, nn_ = first_attn(l + 1), first_norm(l + 1)
                tl_.update(next_in_proj=hip.packed_weight(an.in_proj_weight[:3 * D], an.in_proj_bias[:3 * D], **tkw),
                           next_qfold=hip.packed_weight(an.linear_rpe.weight[:D], None, wt=True, groups=NH, **tkw),
                           next_norm=(nn_.weight, nn_.bias, nn_.eps), qkv_out=qkv_alt, kv16_out=kv16_alt)
            fold_img = (lambda at: hip.packed_weight(at.linear_rpe.weight[D:], at.linear_rpe.bias[D:], groups=NH, mfma32=True)) if tmf else attn_fold_image
            hip.knarpe_dec_mid(qkv, 0, 3 * D, x, self_seg, list(cross(l)), a1.linear_rpe.bias, a2.linear_rpe.bias,
                               (layer.norm1.weight, layer.norm1.bias, layer.norm1.eps), n, S, fold_img(a1),
                               hip.packed_weight(a1.out_proj_weight, a1.out_proj_bias, **tkw),
                               hip.packed_weight(a2.in_proj_weight[:D], a2.in_proj_bias[:D], **tkw),
                               hip.packed_weight(a2.linear_rpe.weight[:D], None, wt=True, groups=NH, **tkw), fold_img(a2),
                               None, None, fxy, fyw, tail=tl_)
            if not last:
                qkv, qkv_alt = qkv_alt, qkv  # the next layer's q | k | v | qt went to the other buffer (this layer's K/V rows were still being read)
                kv16, kv16_alt = kv16_alt, kv16
            elif tail is not None and not tail_in_launch:  # the caller's row-local stages on the finished rows: a short chain of their own
                ch = layer_chain(rows)
                ch.load(x, BUF1, 0, n=D)
                tail(ch)
                ch.run(rows)
            continue
        if mid:
            # one launch: self attention -> x += out_proj(.) -> LN_1 -> q -> W_k^T q -> cross attention -> obuf (128 wide) + flag
            a2 = layer.attn
            hip.knarpe_dec_mid(qkv, 0, 3 * D, x, self_seg, list(cross(l)), a1.linear_rpe.bias, a2.linear_rpe.bias,
                               (layer.norm1.weight, layer.norm1.bias, layer.norm1.eps), n, S, attn_fold_image(a1),
                               hip.packed_weight(a1.out_proj_weight, a1.out_proj_bias, gemv=True),
                               hip.packed_weight(a2.in_proj_weight[:D], a2.in_proj_bias[:D], gemv=True),
                               hip.packed_weight(a2.linear_rpe.weight[:D], None, wt=True, groups=NH, gemv=True), attn_fold_image(a2),
                               obuf, flag, fxy, fyw)
            ch = layer_chain(rows)
            emit_attn_out(ch, a2, obuf, flag, drop=None, x=x)
            if not emit_ffn(ch, layer, zero_rows=src_invalid):
                ch.rowmask(BUF1, 0, D, mask=src_invalid)
            ch.store(BUF1, 0, D, x)
            if l + 1 < len(layers):
                emit_proj(ch, rows, first_norm(l + 1), first_attn(l + 1), qkv, with_kv=True, kv16=kv16)
            elif tail is not None:
                tail(ch)
            ch.run(rows)
            continue
        attention(qkv, 0, 3 * D, a1, n, S, [self_seg], obuf, flag, fxy, fyw, drop=next_call(a1), fold=attn_fold_image(a1) if fold else None)
        if tile:  # the layer's row-local chains as tbx_layer_tile launches (split-bf16 MFMA stages, no program to interpret)
            a_last = a1
            if dec:
                hip.layer_tile(x, attn=tile_attn_part(a1, obuf, flag), proj=tile_proj_part(layer.norm1, layer.attn, q2, False),
                               drop=_tile_drop(next_site()))
                attention(q2, 0, D, layer.attn, n, S, list(cross(l)), obuf, flag, fxy, fyw, drop=next_call(layer.attn))
                a_last = layer.attn
            last = l + 1 == len(layers)
            hip.layer_tile(x, attn=tile_attn_part(a_last, obuf, flag), ffn=tile_ffn_part(layer, src_invalid),
                           proj=None if last else tile_proj_part(first_norm(l + 1), first_attn(l + 1), qkv, True, kv16),
                           drop=_tile_drop(next_site(), next_site(), next_site()))
            if last and tail is not None:  # the caller's row-local stages on the finished rows
                lights = (tail_mf() if callable(tail_mf) else tail_mf) if (current().tl_tail_tile and DROP_CTX is None) else None
                if lights is not None:  # the lights' tail (K/V tables of the agents' layers + next-state logits) as ONE tile launch
                    hip.tl_tail_tile(x, lights)
                else:  # ... or a short row chain of their own
                    ch = layer_chain(rows)
                    ch.load(x, BUF1, 0, n=D)
                    tail(ch)
                    ch.run(rows)
            continue
        ch = layer_chain(rows)
        emit_attn_out(ch, a1, obuf, flag, drop=next_site(), x=x)
        if dec:
            ch.store(BUF1, 0, D, x)
            emit_proj(ch, rows, layer.norm1, layer.attn, q2, with_kv=False)
            ch.run(rows)
            attention(q2, 0, D, layer.attn, n, S, list(cross(l)), obuf, flag, fxy, fyw, drop=next_call(layer.attn),
                      fold=attn_fold_image(layer.attn) if fold else None)
            ch = layer_chain(rows)
            emit_attn_out(ch, layer.attn, obuf, flag, drop=next_site(), x=x)
        if not emit_ffn(ch, layer, drop_hidden=next_site(), drop_out=next_site(), zero_rows=src_invalid):
            ch.rowmask(BUF1, 0, D, mask=src_invalid)
        ch.store(BUF1, 0, D, x)
        if l + 1 < len(layers):
            emit_proj(ch, rows, first_norm(l + 1), first_attn(l + 1), qkv, with_kv=True, kv16=kv16)
        elif tail is not None:
            tail(ch)
        ch.run(rows)
    if drop is not None:
        drop["site"], drop["call"] = DROP_CTX["site"], DROP_CTX["call"]
        DROP_CTX = outer
    return heads_done


def kv_tables(x: torch.Tensor, norms_and_attns, out: Optional[torch.Tensor] = None, tile_rows: int = 16) -> torch.Tensor:
    """Per-token K/V tables of a target set for several attention layers at once:
    out[:, l*256:(l+1)*256] = LN_l(x) @ W_kv,l^T + b_kv,l   (transformer_rpe.py:220-223 + attention_rpe.py:92-98,
    projected before the gather)."""
    rows = x.shape[0]
    L = len(norms_and_attns)
    if out is None:
        out = torch.empty(rows, 2 * D * L, dtype=kv_dtype(), device=x.device)
    ch = row_chain(rows, 132, 132, 132, big=(32, 132, 132, 132))
    ch.load(x, BUF1, 0, n=D)
    emit_kv_tables(ch, norms_and_attns, out)
    ch.run(rows)
    return out


def emit_kv_tables(ch: Chain, norms_and_attns, out: torch.Tensor, x_buf: int = BUF1):
    for l, (nm, attn) in enumerate(norms_and_attns):
        ch.layernorm(x_buf, 0, BUF0, 0, nm.weight, nm.bias, nm.eps)
        ch.linear(BUF0, 0, GLOBAL, l * 2 * D, attn.in_proj_weight[D:], attn.in_proj_bias[D:], out=out)
