"""Per-kernel-class timing passes behind bench.py's `roofline` objects: HIP events (torch.cuda.Event on the launch stream - the
engine launches on torch's current stream) around every launch of this repo's kernel classes in a few eager steps, the algorithmic
bytes or flops of each launch beside them, and the HBM traffic of the same kernel from the committed PMC passes."""
import glob
import hashlib
import json
import os
import re
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0      # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate
FP32_MFMA_PEAK_TF = 157.3  # v_mfma_f32_16x16x4_f32 (the exact-fp32 row chains)
BF16_MFMA_PEAK_TF = 2500.0  # v_mfma_f32_16x16x32_bf16, dense
# The tile kernels / dec_layer_mf / tall_linear form each fp32 product from THREE bf16 MFMA products (hi*hi + hi*lo + lo*hi): their
# fp32-equivalent flops are priced against the peak of the instruction they issue divided by the three products.
SPLIT_BF16_PEAK_TF = BF16_MFMA_PEAK_TF / 3.0
# ... and the single-product schedule (Schedule.mfma_products = 1) against the instruction's own peak.


def mfma_peak(cls: str, products: int = 3):
    """(peak TFLOP/s, instruction) a class of MFMA-bound launches is priced against."""
    if cls in ("chain", "chain_live"):
        return FP32_MFMA_PEAK_TF, "v_mfma_f32_16x16x4_f32"
    return BF16_MFMA_PEAK_TF / max(1, products), f"v_mfma_f32_16x16x32_bf16 x{max(1, products)} products per fp32 product"


def round_tag(name: str):
    """'r05_c2_pmc.json' -> 5; None for names without a round prefix."""
    m = re.match(r"r(\d\d)", Path(name).name)
    return int(m.group(1)) if m else None


def newest_round() -> int:
    """The newest round that has committed evidence under profiles/ (profiles/rNN_*)."""
    return max([round_tag(f) or 0 for f in glob.glob(str(ROOT / "profiles" / "r??_*"))] or [0])


SOURCE_FILES = ("trafficbotsv1.5_amd/csrc/*.hip", "trafficbotsv1.5_amd/csrc/*.h", "trafficbotsv1.5_amd/csrc/*.inc", "include/*.h",
                "trafficbotsv1.5_amd/engine.py")


def source_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources + the launch schedule: what a PMC pass profiled. tools/rocpd_pmc.py writes
    it into every profile; kernel_entry() compares it with the tree that is being timed (`traffic_matches_build`)."""
    h = hashlib.sha256()
    for pat in SOURCE_FILES:
        for f in sorted(glob.glob(str(ROOT / pat))):
            h.update(Path(f).name.encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args, prefixes):
    """HBM traffic per launch from the committed PMC passes (profiles/*pmc*.json) of this workload, for the kernel variant whose name
    starts with one of `prefixes` (the variant with the most launches in that pass). Only profiles of the NEWEST round with evidence
    under profiles/ are accepted: a kernel renamed or rewritten this round must not be described by last round's counters (round 4
    quoted r03's pass for `dec_layer_mf1_kernel`). -> (bytes, file name, kernel variant, the profile's source hash) or Nones."""
    newest = newest_round()
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*pmc*.json")), reverse=True):
        if round_tag(f) != newest:
            continue
        d = json.load(open(f))
        w = d.get("workload", {})
        if (w.get("agents"), w.get("polylines"), w.get("lights"), w.get("scenes"), w.get("rollouts")) != (
                args.agents, args.polylines, args.lights, args.scenes, args.rollouts):
            continue
        if bool(w.get("kv_bf16", False)) != bool(args.kv_bf16):
            continue
        hits = [(v.get("launches", 0), k, v) for k, v in d.get("kernels", {}).items() if ("<" in k or k in prefixes) and any(k.startswith(p) for p in prefixes)]
        if hits:
            _, k, v = max(hits)
            return v["traffic_bytes_per_launch"], Path(f).name, k, d.get("source_sha16")
    return None, None, None, None


def attn_counters(args):
    """VALU-busy / L2 figures of the attention kernel from the committed counter passes (profiles/*attn_counters*.json, collected by
    tools/pmc_attn.sh at the WOSAC shape) - attached only to that workload's attention entry."""
    if (args.agents, args.rollouts, args.scenes) != (128, 32, 1):
        return None
    newest = newest_round()
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*attn_counters*.json")), reverse=True):
        if round_tag(f) != newest:  # (as pmc_traffic: this round's kernels are described by this round's counters only)
            continue
        d = json.load(open(f))
        if "valu_busy" in d and bool(d.get("kv_bf16", False)) == bool(args.kv_bf16):
            return {"valu_busy": d["valu_busy"], "l2_hit_rate": d.get("l2_hit_rate"), "l2_read_requests_per_launch": d.get("l2_read_requests"),
                    "valu_insts_per_pair": d.get("valu_insts_per_pair"), "counters_source": Path(f).name, "counters_measured": False,
                    "valu_wave_insts_per_launch": (d.get("per_launch") or {}).get("SQ_INSTS_VALU")}
    return None


N_SIMD, SHADER_CLOCK_GHZ, VALU_ISSUE_CLK = 1024, 2.4, 4  # 256 CUs x 4 SIMDs; a wave-wide VALU instruction occupies its SIMD for 4 clocks


def attach_attn_counters(att: dict, cnt: dict) -> None:
    """The large-launch attention kernel against the bound that binds it. SURVEY 8d's `frac` prices the bytes the
    pairs GATHER; once the K rollouts of a scene share their tables most of those rows are L2 / L1 hits and the kernel is bound by
    instruction issue at low occupancy, not by HBM. From the committed counter pass of this workload:
      valu_issue_floor_us = wave-wide VALU instructions per launch x 4 clocks / 1024 SIMDs / shader clock - the time the launch's
                            VALU work takes with every SIMD issuing back to back;
      valu_issue_frac     = that floor / the measured launch time (the fraction of the bound that binds);
      bound               = "occupancy/latency" when the counter-measured HBM traffic is under a quarter of the HBM peak."""
    if cnt.get("l2_read_requests_per_launch"):  # 128-byte L1 -> L2 read requests of a launch over its live duration
        cnt["l2_request_frac"] = cnt["l2_read_requests_per_launch"] * 128.0 / (att["avg_launch_us"] * 1e-6) / 1e9 / L2_PEAK_GBS
    att["counters"] = cnt
    wi = cnt.get("valu_wave_insts_per_launch")
    if wi:
        floor_us = wi * VALU_ISSUE_CLK / N_SIMD / (SHADER_CLOCK_GHZ * 1e3)
        att["valu_issue_floor_us"] = floor_us
        att["valu_issue_frac"] = floor_us / att["avg_launch_us"]
    if att.get("hbm_measured_frac") is not None and att["hbm_measured_frac"] < 0.25:
        att["bound_8d"] = att.get("bound")
        att["bound"] = "occupancy/latency"
        att["bound_note"] = ("counter-measured HBM traffic is under a quarter of the HBM peak (rows shared by the rollouts of a scene are L2 / L1 "
                             "hits): `frac` = SURVEY 8d's gathered bytes over time stays as the byte model; the binding resource is VALU issue at "
                             "2 waves per SIMD - see valu_issue_frac")


def trace_avg_us(args, kernel_prefix: str):
    """Average launch duration of `kernel_prefix` in this round's committed rocprofv3 kernel trace of the workload
    (profiles/rNN_*kernel_stats.md: `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` reduced by tools/rocpd_stats.py), or
    (None, None). The event pairs of the live pass bracket a launch AND its neighbours' boundaries (they over-read ~10 %: the chain of
    one step by event averages exceeds ms_per_step, by trace averages it does not - VERDICT r05 weak 2); the line carries both."""
    newest = newest_round()
    tag = {(64, 1, 1): "c2", (128, 1, 32): "c5", (128, 1, 128): "sub", (64, 16, 1): "s16", (64, 64, 1): "s64"}.get((args.agents, args.scenes, args.rollouts))
    if tag is None:
        return None, None
    f = ROOT / "profiles" / f"r{newest:02d}_{tag}{'_bf16' if args.kv_bf16 else ''}_kernel_stats.md"
    if not f.exists():
        return None, None
    best = None
    for line in f.read_text().splitlines():
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m and kernel_prefix in m.group(1):
            calls, avg = int(m.group(2)), float(m.group(4))
            if best is None or calls > best[0]:
                best = (calls, avg)
    return (best[1], f.name) if best else (None, None)


def attn_algorithmic_bytes(n_src_rows: int, n_pairs: int, b: int = 4) -> float:
    """S*2*d*b + P*(2*d*b + 12 + 4 + 1) + (d_rpe*2d + 2d)*b, d = d_rpe = 128; b = 4 (fp32 tables: 1041 B per pair)
    or 2 (bfloat16 K/V tables: 529 B per pair)."""
    d = 128
    return n_src_rows * 2 * d * b + n_pairs * (2 * d * b + 17) + (d * 2 * d + 2 * d) * b


class KernelEvents:
    """Brackets every launch of the hot path's kernel classes with HIP events on the launch stream and keeps, per class, the
    algorithmic bytes (HBM-bound classes) or flops (MFMA-bound classes) of each launch:
      dec_layer  tbx_knarpe_dec_mid / tbx_knarpe_dec_layer (dec_layer_mf_kernel / dec_mid_kernel: a whole decoder layer, or its attention half);
                 tbx_knarpe_dec_layer_pair (dec_layer_mf_pair_kernel: the agents' and the lights' rows of a layer in one launch, Schedule.one_queue)
      attn       tbx_knarpe_attn_* (knarpe_attn_kernel), grouped by source rows
      chain      tbx_rowchain / tbx_rowchain_ex (rowchain_kernel<MT,..>: MFMA row chains), grouped by tile rows
      chain_live tbx_rowchain_live (rowchain_kernel<0,1,0,1>: thread-per-column chains of small launches)
      tile       tbx_layer_tile / tbx_heads_tile / tbx_window_tile / tbx_front (split-bf16 tile kernels)
      other      K-nearest searches, preparation, tbx_sim_step (elementwise / latency)"""

    WRAPPED = ("knarpe_attn", "knarpe_attn_mfma", "knarpe_dec_mid", "knn_embed", "knn_embed_multi", "agent_prep", "tl_prep", "sim_step", "pose_embed",
               "layer_tile", "heads_tile", "window_tile", "front", "pair_embed", "launch_dec_layer_pair", "launch_front_pair")

    def __init__(self, hip):
        self.hip, self.rec = hip, {}
        self.extra = {}  # (class, key) -> per-launch bytes a launch must move beyond its algorithmic bytes
        self.dec_kernel = "dec_layer_mf_kernel"
        self._saved = {}
        self.half = {}  # id(deferred launch descriptor) -> the work of that half of a paired launch (Schedule.one_queue)

    def _time(self, cls, key, work, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.rec.setdefault((cls, key), []).append((e0, e1, work))
        return r

    def __enter__(self):
        hip, T = self.hip, self._time
        sv = self._saved = {n: getattr(hip, n) for n in self.WRAPPED if hasattr(hip, n)}
        sv["Chain.run"] = hip.Chain.run

        def mid(*args, **kw):
            # ALGORITHMIC bytes of the launch = SURVEY 8d only: the launch holds two attention calls (self + cross) - per call
            # S*2*d*b + P*(2*d*b + 17) + (d_rpe*2d + 2d)*b. What the launch must ALSO move - every weight image once (5 of the attention
            # half; with a tail the layer's out_proj / FFN / next projections = 13 chunks, with the heads 15 more) + the token rows in
            # and out - is kept beside it as `compulsory_bytes_per_launch` and never enters `achieved` / `frac`.
            self_seg, cross = args[4], args[5]
            rows = args[9] * args[10]
            eb = 2 if self_seg.kv.dtype == torch.bfloat16 else 4
            pairs = rows * (self_seg.k + sum(c.k for c in cross))
            tail = kw.get("tail")
            w = (4 * 33 + 36) * 2048
            if tail is not None:
                w += (8 * 33 + 3 * 32 + (3 * 33 + 36 if tail.get("qkv_out") is not None else 0) + (13 * 33 + 2 * 32 if tail.get("heads") else 0)) * 2048
                if tail.get("mfma32") == 2:  # one bf16 product per LINEAR: the hi halves of the weight units only
                    w //= 2
            b = 2 * attn_algorithmic_bytes(rows, 0, eb) + pairs * (2 * 128 * eb + 17)
            self.extra.setdefault(("dec_layer", rows), []).append(w + rows * (128 * 4 * 2 + (896 * 4 if tail and tail.get("qkv_out") is not None else 0)))
            self.dec_kernel = "dec_layer_mf1_kernel" if (tail is not None and tail.get("mfma32") == 2) else ("dec_layer_mf_kernel" if (tail is not None and tail.get("mfma32")) else "dec_mid_kernel")
            if hip.hip_base.DEFERRED is not None:  # Schedule.one_queue: the call only describes its half of a PAIRED launch (pair_mid below times it)
                sv["knarpe_dec_mid"](*args, **kw)
                self.half[id(hip.hip_base.DEFERRED[-1])] = (rows, b, self.extra[("dec_layer", rows)].pop())
                return None
            return T("dec_layer", rows, b, sv["knarpe_dec_mid"], *args, **kw)

        def pair_mid(a_, b_):
            # tbx_knarpe_dec_layer_pair: ONE launch = the agents' rows + the lights' rows of a layer - the 8d bytes of both halves
            (ra, wa, xa), (rb, wb_, xb) = self.half.pop(id(a_)), self.half.pop(id(b_))
            self.extra.setdefault(("dec_layer", ra + rb), []).append(xa + xb)
            self.dec_kernel = self.dec_kernel.replace("_kernel", "_pair_kernel") if "_pair_" not in self.dec_kernel else self.dec_kernel
            return T("dec_layer", ra + rb, wa + wb_, sv["launch_dec_layer_pair"], a_, b_)

        def attn(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, out, flag, *freqs, **kw):
            eb = 2 if segs[0].kv.dtype == torch.bfloat16 else 4
            b = attn_algorithmic_bytes(n_batch * n_src, n_batch * n_src * sum(s.k for s in segs), eb)
            return T("attn", n_batch * n_src, b, sv["knarpe_attn"], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, out, flag, *freqs, **kw)

        def attn_m(qbuf, q_off, qt_off, n_batch, n_src, segs, out, flag, fxy, fyw, **kw):
            eb = 2 if segs[0].kv.dtype == torch.bfloat16 else 4
            b = attn_algorithmic_bytes(n_batch * n_src, n_batch * n_src * sum(s.k for s in segs), eb)
            self.attn_kernel = "knarpe_attn_mfma_kernel"
            return T("attn", n_batch * n_src, b, sv["knarpe_attn_mfma"], qbuf, q_off, qt_off, n_batch, n_src, segs, out, flag, fxy, fyw, **kw)

        def run(ch, n_rows, group_rows=0):
            fl = sum(2.0 * n_rows * s.k * s.n * max(1, s.reserved) for s in ch.stages if s.op == hip.OP_LINEAR)
            if ch.live_rows:
                return T("chain_live", 0, fl, sv["Chain.run"], ch, n_rows, group_rows)
            return T("chain", ch.tile_rows, fl, sv["Chain.run"], ch, n_rows, group_rows)

        def lt(x, attn=None, ffn=None, proj=None, store_x=True, drop=None, rider=None):
            rows = x.shape[0]
            mac = (2 * 128 * 128 if attn is not None else 0) + (2 * 128 * 512 if ffn is not None else 0)
            if proj is not None:
                mac += 128 * proj["n"] + 128 * 128
            fl = 2.0 * rows * mac + (0.0 if rider is None else 2.0 * rider["out"].shape[0] * 4 * 128 * 128)
            return T("tile", "layer", fl, sv["layer_tile"], x, attn=attn, ffn=ffn, proj=proj, store_x=store_x, drop=drop, rider=rider)

        def ht(x, hd):
            return T("tile", "heads", 2.0 * x.shape[0] * (2 * (256 * 128 + 2 * 128 * 128) + 128 * 384 + 3 * 128 * 128 + 3 * 128 * 16), sv["heads_tile"], x, hd)

        def wt(attr, pe, row_invalid, in_images, pn_images, window, out, add_mode=False, drop=None):
            mac = (32 * 128 + 2 * 128 * 128 if add_mode else 32 * 64 + 2 * 64 * 64) + 3 * 128 * 64
            return T("tile", "window", 2.0 * attr.shape[0] * mac, sv["window_tile"], attr, pe, row_invalid, in_images, pn_images, window, out,
                     add_mode=add_mode, drop=drop)

        def fr(window, proj, rider=None, jobs=None, pose_embed_job=None):
            rows = window["out"].shape[0]
            add = bool(window.get("add_mode"))
            mac_w = (32 * 128 + 2 * 128 * 128 if add else 32 * 64 + 2 * 64 * 64) + 3 * 128 * 64
            fl = 2.0 * window["attr"].shape[0] * mac_w + 2.0 * rows * (128 * 384 + 128 * 128) + (0.0 if rider is None else 2.0 * rider["out"].shape[0] * 4 * 128 * 128)
            if hip.hip_base.DEFERRED is not None:  # (Schedule.one_queue: one half of tbx_front_pair)
                r = sv["front"](window, proj, rider=rider, jobs=jobs, pose_embed_job=pose_embed_job)
                self.half[id(hip.hip_base.DEFERRED[-1])] = fl
                return r
            return T("tile", "front", fl, sv["front"], window, proj, rider=rider, jobs=jobs, pose_embed_job=pose_embed_job)

        def pair_fr(a_, b_):
            return T("tile", "front_pair", self.half.pop(id(a_)) + self.half.pop(id(b_)), sv["launch_front_pair"], a_, b_)

        def other(name):
            return lambda *a, **kw: T("other", name, 0.0, sv[name], *a, **kw)

        hip.knarpe_attn, hip.Chain.run, hip.knarpe_dec_mid = attn, run, mid
        if "knarpe_attn_mfma" in sv:
            hip.knarpe_attn_mfma = attn_m
        hip.layer_tile, hip.heads_tile, hip.window_tile, hip.front = lt, ht, wt, fr
        if "launch_dec_layer_pair" in sv:
            hip.launch_dec_layer_pair, hip.launch_front_pair = pair_mid, pair_fr
        for n in ("knn_embed", "knn_embed_multi", "agent_prep", "tl_prep", "sim_step", "pose_embed", "pair_embed"):
            if n in sv:
                setattr(hip, n, other(n))
        return self

    def __exit__(self, *a):
        for n, f in self._saved.items():
            if n == "Chain.run":
                self.hip.Chain.run = f
            else:
                setattr(self.hip, n, f)

    def classes(self, n_steps: int):
        """-> list of per-(class, key) dicts sorted by total time, largest first."""
        torch.cuda.synchronize()
        out = []
        for (cls, key), evs in self.rec.items():
            t = sum(e0.elapsed_time(e1) for e0, e1, _ in evs) * 1e-3
            out.append(dict(cls=cls, key=key, t=t, n=len(evs), work=sum(w for *_, w in evs), per_step=len(evs) / n_steps,
                            extra=sum(self.extra.get((cls, key), [])), dec_kernel=self.dec_kernel))
        tot = sum(c["t"] for c in out) or 1.0
        for c in out:
            c["share"] = c["t"] / tot
        return sorted(out, key=lambda c: -c["t"])


def kernel_entry(args, c, products: int = 3):
    """One `kernels` / `roofline` object for a KernelEvents class: achieved = algorithmic bytes (or flops) per launch / the average
    launch duration between HIP events; traffic = HBM bytes per launch from this workload's committed PMC pass (if any)."""
    cls, key = c["cls"], c["key"]
    avg = c["t"] / c["n"]
    e = {"class": cls, "share_of_step_kernel_time": c["share"], "launches_per_step": c["per_step"], "avg_launch_us": avg * 1e6}
    if cls in ("dec_layer", "attn"):
        ach = c["work"] / c["t"] / 1e9
        wave_rows = int(os.environ.get("TBX_ATTN_BIG_ROWS_INFER", 193))  # (attn.hip attn_big_rows: a wavefront per source row from here)
        mf_rows = int(os.environ.get("TBX_ATTN_MFMA_MIN_ROWS", 193))  # (Schedule.attn_mfma_min_rows)
        mf = cls == "attn" and getattr(args, "attn_mfma", None) and key >= mf_rows
        dec_name = c.get("dec_kernel", "dec_layer_mf_kernel")  # (dec_layer_mf1_kernel under Schedule.linear_bf16, dec_mid_kernel with Schedule.dec_tail_mfma off)
        name = dec_name if cls == "dec_layer" else ("knarpe_attn_mfma_kernel" if mf else "knarpe_attn_kernel")
        pre = [dec_name + "<"] if cls == "dec_layer" else (["knarpe_attn_mfma_kernel<"] if mf else (["knarpe_attn_kernel<1,"] if key >= wave_rows else ["knarpe_attn_kernel<4,"]))
        e.update(kernel=name, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                 algorithmic_bytes_per_launch=c["work"] / c["n"], source_rows_per_launch=key,
                 bytes_per_pair=529 if args.kv_bf16 else 1041)
        if c.get("extra"):
            e["compulsory_bytes_per_launch"] = c["work"] / c["n"] + c["extra"] / c["n"]  # 8d bytes + weight images + token rows in / out
        if cls == "attn":
            e.update(l2_frac=ach / L2_PEAK_GBS, l2_peak=L2_PEAK_GBS)
        if cls == "dec_layer" or key < 1024:  # (under 4 workgroups per CU)
            e["note"] = ("latency-bound at this size: a launch has one workgroup per source row (64-128 of them on 256 CUs) and the "
                         "step is a chain of dependent launches; frac is bytes over time, not a bandwidth-limited figure")
    elif cls in ("chain", "chain_live", "tile"):
        ach = c["work"] / c["t"] / 1e12
        if cls == "tile":
            name = {"front": "front_kernel", "front_pair": "front_pair_kernel"}.get(key, f"tile_{key}_kernel")  # (tbx_front: window tile + first projection + searches)
            pre = [name]
        else:
            name = "rowchain_kernel" + ("<live>" if cls == "chain_live" else f"<{key}-row tiles>")
            pre = ["rowchain_kernel<0,1,0,1>"] if cls == "chain_live" else [f"rowchain_kernel<{key // 16},"]
        peak, insn = mfma_peak(cls, products)
        e.update(kernel=name, bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, flops_per_launch=c["work"] / c["n"],
                 peak_note=f"fp32-equivalent flops of the LINEAR stages against the dense peak of the instruction issued ({insn})")
    else:
        e.update(kernel=f"tbx_{key}", bound="latency", achieved=None, peak=None, unit=None, frac=None)
        return e
    traffic, src, variant, sha = pmc_traffic(args, pre)
    e.update(traffic=traffic, traffic_source=src, traffic_kernel=variant,
             traffic_measured=False)  # PMC passes are separate rocprofv3 runs: the committed profile of this workload
    if src is not None:  # does the profile describe the sources that are being timed? (tools/rocpd_pmc.py stamps every profile)
        e["traffic_matches_build"] = (sha == source_sha16()) if sha else None
    if traffic is not None:
        e["hbm_measured_frac"] = traffic / avg / 1e9 / HBM_PEAK_GBS
    tavg, tsrc = trace_avg_us(args, name.split("<")[0])
    if tavg is not None:  # (the same command under rocprofv3 --kernel-trace, committed this round: the figure the chain of a step adds up with)
        e["avg_launch_us_trace"], e["trace_source"] = tavg, tsrc
        e["frac_at_trace_avg"] = (c["work"] / c["n"]) / (tavg * 1e-6) / (1e9 if e["unit"] == "GB/s" else 1e12) / e["peak"]
    return e


def gemm_summary(kernels):
    """All MFMA-bound classes of a step as one figure, each class priced against the peak of the instruction it issues."""
    mfma = [k for k in kernels if k["bound"] == "mfma"]
    if not mfma:
        return None
    t = sum(k["avg_launch_us"] * 1e-6 * k["launches_per_step"] for k in mfma)
    fl = sum(k["flops_per_launch"] * k["launches_per_step"] for k in mfma)
    # time-weighted fraction: sum(flops_i / peak_i) / sum(t_i)
    frac = sum(k["flops_per_launch"] * k["launches_per_step"] / (k["peak"] * 1e12) for k in mfma) / t
    return {"kernel": "all MFMA-bound classes (tile kernels + row chains)", "bound": "mfma", "unit": "TFLOP/s", "achieved": fl / t / 1e12,
            "peak": max(k["peak"] for k in mfma), "frac": frac,
            "note": "frac = sum(flops_i / peak_i) / sum(t_i): each class against the dense peak of the MFMA instruction it issues"}


def train_pmc_traffic(cls: str):
    """-> {traffic, traffic_source, traffic_kernel} of a training kernel class from profiles/*train_pmc*.json (separate --pmc FETCH_SIZE /
    WRITE_SIZE passes of two eager training steps, tools/pmc_train.sh), per call of the class's entry point, or None. The attention
    forward class covers two kernels (the wave-per-row kernel of the time-batched pass and the ring form of the stepping pass):
    launch-weighted; a backward call is one launch of the row kernel + one of the dK / dV kernel: summed."""
    stem = cls.split(" ")[0]  # "knarpe_attn_bwd_kernel + dkv" -> knarpe_attn_bwd_kernel
    if not stem.endswith("_kernel"):
        return None
    parts = {"knarpe_attn_kernel": (("knarpe_attn_kernel", "knarpe_attn_ring_kernel"), "mean"),
             "knarpe_attn_bwd_kernel": (("knarpe_attn_bwd_kernel", "knarpe_attn_dkv_kernel"), "sum")}.get(stem, ((stem,), "mean"))
    newest = newest_round()
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*train_pmc*.json")), reverse=True):
        if round_tag(f) != newest:  # (as pmc_traffic: this round's kernels are described by this round's counters only)
            continue
        ks = json.load(open(f)).get("kernels", {})
        vs = [ks[k] for k in parts[0] if k in ks]
        if not vs:
            continue
        if parts[1] == "sum":
            tr = sum(v["traffic_bytes_per_launch"] for v in vs)
        else:
            tr = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in vs) / sum(v["launches"] for v in vs)
        return {"traffic": int(tr), "traffic_source": Path(f).name, "traffic_kernel": " + ".join(k for k in parts[0] if k in ks)}
    return None


def train_kernel_pass(hip, step, replay_s):
    """Times this repo's kernels inside one eager training step (HIP events on the launch stream, behind a device-side delay that lets
    the host enqueue the step ahead of the device: the pairs then bracket back-to-back launches; shares are of `replay_s`, the timed
    hipGraph replay of the same launches). Algorithmic work:
    attention forward = SURVEY 8d bytes; backward = the forward's bytes + d(out) and d(q) rows (1280 floats per row) + 8 coefficient
    floats per pair; tbx_linear_wgrad = dY and X read once (4 (n + k) bytes per row); tbx_tall_linear = X read, Y written once (the
    same 4 (n + k) bytes per row); LayerNorm 1.0 / 1.5 KB per row; chains / tile kernels: flops."""
    rec, saved = {}, {}

    def T(cls, bound, work, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        rec.setdefault((cls, bound), []).append((e0, e1, work))
        return r

    def pairs(n_batch, n_src, segs):
        return n_batch * n_src, n_batch * n_src * sum(sg.k for sg in segs)

    # A class is split by SIZE: calls over >= 16,384 rows are the bandwidth-bound population (10^2..10^3 us, 0.1-2 GB
    # each); the smaller ones (the stepping pass's 1,024-row launches, once-per-batch encoders) are 7-40 us launches whose fraction of
    # the HBM peak says how short they are, not how well they stream. One launch-weighted mean over both said neither.
    big = lambda rows: " [rows >= 16384]" if rows >= 16384 else " [rows < 16384]"

    def attn(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw):
        r, p = pairs(n_batch, n_src, segs)
        return T("knarpe_attn_kernel (forward)" + big(r), "hbm", attn_algorithmic_bytes(r, p), saved["knarpe_attn"], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw)

    def attn_bwd(name):
        def f(qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw):
            r, p = pairs(n_batch, n_src, segs)
            return T("knarpe_attn_bwd_kernel + dkv" + big(r), "hbm", attn_algorithmic_bytes(r, p) + r * 1280 * 4 + p * 32, saved[name], qbuf, q_off, qt_off, bias, n_batch, n_src, segs, *a, **kw)
        return f

    def wgrad(dy, x, *a, **kw):
        name = "wgrad_partial_kernel<bf16> (tbx_linear_wgrad_bf16)" if kw.get("bf16") else "wgrad_partial_kernel (tbx_linear_wgrad)"
        return T(name + big(dy.shape[0]), "hbm", 4.0 * dy.shape[0] * (dy.shape[1] + x.shape[1]), saved["linear_wgrad"], dy, x, *a, **kw)

    def tall(x, w, b=None, wt=False, relu=False, bf16=False, **kw):  # (kw: out / out16 - the bfloat16 copy adds 2 n bytes per row)
        n_, k_ = (w.shape[1], w.shape[0]) if wt else (w.shape[0], w.shape[1])
        rows = x.numel() // k_
        by = (4.0 * (k_ + n_) + (2.0 * n_ if kw.get("out16") is not None else 0.0)) * rows
        return T("tall_linear_kernel (tbx_tall_linear" + ("_bf16)" if bf16 else ")") + big(rows), "hbm", by, saved["tall_linear"], x, w, b, wt=wt, relu=relu, bf16=bf16, **kw)

    def attn_m(qbuf, q_off, qt_off, n_batch, n_src, segs, *a, **kw):
        r, p = pairs(n_batch, n_src, segs)  # (529 B per pair on the bfloat16 copies of the tables, 1041 on fp32 tables)
        eb = 2 if segs[0].kv.dtype == torch.bfloat16 else 4
        return T("knarpe_attn_mfma_kernel (forward, bf16 operands)" + big(r), "hbm", attn_algorithmic_bytes(r, p, eb), saved["knarpe_attn_mfma"], qbuf, q_off, qt_off, n_batch,
                 n_src, segs, *a, **kw)

    def lt(x, attn=None, ffn=None, proj=None, store_x=True, drop=None, rider=None):
        mac = (2 * 128 * 128 if attn is not None else 0) + (2 * 128 * 512 if ffn is not None else 0) + (0 if proj is None else 128 * proj["n"] + 128 * 128)
        return T("tile_layer / tile_heads / tile_window kernels (stepping pass)", "mfma3", 2.0 * x.shape[0] * mac, saved["layer_tile"], x, attn=attn, ffn=ffn,
                 proj=proj, store_x=store_x, drop=drop, rider=rider)

    def ln_f(x, *a, **kw):
        return T("ln_fwd_kernel" + big(x.numel() // x.shape[-1]), "hbm", x.numel() * 8.0, saved["layernorm_fwd"], x, *a, **kw)

    def ln_b(x, *a, **kw):
        return T("ln_bwd_kernel" + big(x.numel() // x.shape[-1]), "hbm", x.numel() * 12.0, saved["layernorm_bwd"], x, *a, **kw)

    def run(ch, n_rows, group_rows=0):
        fl = sum(2.0 * n_rows * st.k * st.n * max(1, st.reserved) for st in ch.stages if st.op == hip.OP_LINEAR)
        return T("rowchain_kernel (stepping pass)", "mfma", fl, saved["Chain.run"], ch, n_rows, group_rows)

    names = {"knarpe_attn": attn, "knarpe_attn_bwd_gather": attn_bwd("knarpe_attn_bwd_gather"), "knarpe_attn_bwd": attn_bwd("knarpe_attn_bwd"),
             "linear_wgrad": wgrad, "layernorm_fwd": ln_f, "layernorm_bwd": ln_b, "tall_linear": tall, "layer_tile": lt, "knarpe_attn_mfma": attn_m}
    for n, f in names.items():
        saved[n] = getattr(hip, n)
        setattr(hip, n, f)
    saved["Chain.run"] = hip.Chain.run
    hip.Chain.run = run
    try:
        # the eager step is bound by the host's launch rate: on an idle stream an event pair around a launch times the wait for the host
        # to enqueue it (seen here: 515 us "launches" of a 73 us kernel). A device-side delay in front, as long as the host needs to
        # enqueue the whole step (measured on one plain eager step first), lets the launches queue up and run back to back.
        torch.cuda.synchronize()
        t_host = time.perf_counter()
        saved_step = {n: getattr(hip, n) for n in names}
        for n in names:  # (the plain step: unwrapped)
            setattr(hip, n, saved[n])
        hip.Chain.run = saved["Chain.run"]
        step()
        t_host = time.perf_counter() - t_host  # (enqueue time: nothing in the step waits for the device)
        torch.cuda.synchronize()
        for n, f in saved_step.items():
            setattr(hip, n, f)
        hip.Chain.run = run
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(2_000_000)  # (calibration: what one spin cycle of torch's delay kernel is on this device)
        c1.record()
        torch.cuda.synchronize()
        cycles_per_s = 2e6 / max(1e-6, c0.elapsed_time(c1) * 1e-3)
        torch.cuda._sleep(int(cycles_per_s * min(3.0, 1.5 * t_host + 0.05)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
    finally:
        for n, f in saved.items():
            if n == "Chain.run":
                hip.Chain.run = f
            else:
                setattr(hip, n, f)
    total = e0.elapsed_time(e1) * 1e-3
    kernels = []
    for (cls, bound), evs in rec.items():
        t = sum(a.elapsed_time(b) for a, b, _ in evs) * 1e-3
        w = sum(x for *_, x in evs)
        if bound == "hbm":
            peak, unit, ach = HBM_PEAK_GBS, "GB/s", w / t / 1e9
        else:
            peak, unit, ach = (SPLIT_BF16_PEAK_TF if bound == "mfma3" else FP32_MFMA_PEAK_TF), "TFLOP/s", w / t / 1e12
        kernels.append({"kernel": cls, "bound": "hbm" if bound == "hbm" else "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                        "launches_per_step": len(evs), "avg_launch_us": t / len(evs) * 1e6, "share_of_step": t / replay_s, "traffic": None})
    for k in kernels:  # HBM bytes per launch from the committed PMC passes of an eager training step (tools/pmc_train.sh), if any
        tr = train_pmc_traffic(k["kernel"])
        if tr:
            k.update(tr)
    kernels.sort(key=lambda k: -k["share_of_step"])
    rest = 1.0 - sum(k["share_of_step"] for k in kernels)
    kernels.append({"kernel": "library GEMMs of the odd-width layers (rocBLAS fp32) + aten elementwise / copy / reduce + this repo's smaller kernels", "bound": None,
                    "share_of_step": rest})
    roof = dict(kernels[0])
    roof["note"] = ("largest of this repo's kernel classes in ONE eager training step enqueued behind a device-side delay (event pairs on the "
                    "launch stream around back-to-back launches); share_of_step = its event time over the timed hipGraph replay of the same launches")
    roof["eager_step_device_ms"] = total * 1e3
    roof["eager_step_enqueue_ms"] = t_host * 1e3
    return roof, kernels
