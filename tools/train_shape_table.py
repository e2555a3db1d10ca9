"""Per-SHAPE timing of this repo's streaming kernels inside ONE eager training step of configs[2] (16 scenes of 64 / 1024 / 128):
tbx_tall_linear(_bf16), tbx_linear_wgrad(_bf16), LayerNorm forward / backward, the attention forward / backward - HIP events around
every call behind a device-side delay (tools/benchlib/events.py's method), grouped by (kernel, rows, k, n): calls, total ms, average
us, algorithmic bytes per call and the HBM fraction of each group. The class-level fractions on the bench line are launch-weighted
means over calls from 7 us to 2 ms; this table says which calls are slow.   python tools/train_shape_table.py [fp32|bf16]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, '.')
import torch
from importlib import import_module
from __graft_entry__ import load_package
tb = load_package()
W = import_module("trafficbots_amd.pl_modules.waymo_motion")
DP = import_module("trafficbots_amd.pl_modules.data_parallel")
hip = import_module("trafficbots_amd.hip")
torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = W.WaymoMotion(model=tb.config.default_model_cfg(), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg()).to(dev).train()
wm.train_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
(opt,), _ = wm.configure_optimizers()
batch = {k: v.to(dev) for k, v in tb.synthetic.make_scene(16, 64, 1024, 128, seed=0).items()}
step = lambda: DP.train_step(wm, opt, {k: v.clone() for k, v in batch.items()})
step(); step()
torch.cuda.synchronize()
rec, saved = {}, {}


def T(key, nbytes, fn, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **kw)
    e1.record()
    rec.setdefault(key, []).append((e0, e1, nbytes))
    return r


def tall(x, w, b=None, wt=False, relu=False, bf16=False, **kw):
    n_, k_ = (w.shape[1], w.shape[0]) if wt else (w.shape[0], w.shape[1])
    rows = x.numel() // k_
    by = (4.0 * (k_ + n_) + (2.0 * n_ if kw.get("out16") is not None else 0.0)) * rows
    return T(("tall_linear" + ("_bf16" if bf16 else "") + ("+y16" if kw.get("out16") is not None else "") + (" wt" if wt else ""), rows, k_, n_), by,
             saved["tall_linear"], x, w, b, wt=wt, relu=relu, bf16=bf16, **kw)


def wgrad(dy, x, *a, **kw):
    return T(("linear_wgrad" + ("_bf16" if kw.get("bf16") else ""), dy.shape[0], x.shape[1], dy.shape[1]), 4.0 * dy.shape[0] * (dy.shape[1] + x.shape[1]),
             saved["linear_wgrad"], dy, x, *a, **kw)


def ln_f(x, *a, **kw):
    return T(("layernorm_fwd", x.numel() // x.shape[-1], x.shape[-1], x.shape[-1]), x.numel() * 8.0, saved["layernorm_fwd"], x, *a, **kw)


def ln_b(x, *a, **kw):
    return T(("layernorm_bwd", x.numel() // x.shape[-1], x.shape[-1], x.shape[-1]), x.numel() * 12.0, saved["layernorm_bwd"], x, *a, **kw)


E = import_module("tools.benchlib.events")


def attn_f(name, tag, bwd=False):
    def f(qbuf, q_off, qt_off, *a, **kw):
        if name == "knarpe_attn_mfma":
            n_batch, n_src, segs = a[0], a[1], a[2]
        else:
            n_batch, n_src, segs = a[1], a[2], a[3]
        r, p = n_batch * n_src, n_batch * n_src * sum(sg.k for sg in segs)
        eb = 2 if segs[0].kv.dtype == torch.bfloat16 else 4
        by = E.attn_algorithmic_bytes(r, p, eb) + ((r * 1280 * 4 + p * 32) if bwd else 0)
        return T((tag, r, sum(sg.k for sg in segs), eb), by, saved[name], qbuf, q_off, qt_off, *a, **kw)
    return f


names = {"tall_linear": tall, "linear_wgrad": wgrad, "layernorm_fwd": ln_f, "layernorm_bwd": ln_b, "knarpe_attn": attn_f("knarpe_attn", "attn_fwd(valu)"),
         "knarpe_attn_mfma": attn_f("knarpe_attn_mfma", "attn_fwd(mfma)"), "knarpe_attn_bwd": attn_f("knarpe_attn_bwd", "attn_bwd", True),
         "knarpe_attn_bwd_gather": attn_f("knarpe_attn_bwd_gather", "attn_bwd_gather", True)}
for n, f in names.items():
    saved[n] = getattr(hip, n)
    setattr(hip, n, f)
t0 = time.perf_counter(); step(); t_host = time.perf_counter() - t0
torch.cuda.synchronize()
rec.clear()
c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
c0.record(); torch.cuda._sleep(2_000_000); c1.record(); torch.cuda.synchronize()
cps = 2e6 / max(1e-6, c0.elapsed_time(c1) * 1e-3)
torch.cuda._sleep(int(cps * min(3.0, 1.5 * t_host + 0.05)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); step(); e1.record(); torch.cuda.synchronize()
for n, f in saved.items():
    setattr(hip, n, f)
total = e0.elapsed_time(e1)
rows = []
for key, evs in rec.items():
    t = sum(a.elapsed_time(b) for a, b, _ in evs)
    by = sum(x for *_, x in evs)
    rows.append((t, key, len(evs), by))
rows.sort(reverse=True)
print(f"eager step behind a delay: {total:.1f} ms of device time ({wm.train_precision} class); wrapped kernels {sum(r[0] for r in rows):.1f} ms")
print(f"{'kernel':34s} {'rows':>9s} {'k':>5s} {'n':>5s} {'calls':>5s} {'total ms':>9s} {'avg us':>9s} {'MB/call':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
cls = {}
for t, key, n, by in rows:
    gbs = by / (t * 1e-3) / 1e9
    print(f"{key[0]:34s} {key[1]:9d} {key[2]:5d} {key[3]:5d} {n:5d} {t:9.3f} {t / n * 1e3:9.1f} {by / n / 1e6:9.2f} {gbs:8.0f} {gbs / 8000:9.3f}")
    big = "rows >= 16384" if key[1] >= 16384 else "rows < 16384"
    c = cls.setdefault((key[0].split("+")[0].split(" ")[0], big), [0.0, 0.0, 0])
    c[0] += t; c[1] += by; c[2] += n
print()
for (k, big), (t, by, n) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:24s} {big:14s} calls {n:5d}  {t:8.3f} ms  {by / (t * 1e-3) / 1e9 / 8000:6.3f} of HBM peak")
