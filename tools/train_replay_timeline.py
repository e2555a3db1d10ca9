#!/usr/bin/env python
"""Where a hipGraph-replayed training step (bench.py --mode train) spends its wall time, from a rocprofv3 --kernel-trace rocpd database:
the LAST replay of the run is cut at the gaps between steps (the gradient exchange + AdamW run eagerly between two replays), then
  * span, union-busy time (some kernel running), idle time, summed kernel time (concurrency = sum / busy);
  * the step split into PHASES by marker kernels (first stepping-pass launch, last train_chain_fwd launch of the stepping pass, first
    backward kernel), each with span / busy / idle / launches / mean launch;
  * per kernel class inside each phase: launches, summed time, share;
  * the idle gaps: how many, how long, which kernels follow the longest ones.
usage: train_replay_timeline.py <db> [n_top]"""
import collections
import re
import sqlite3
import sys


def short(n: str) -> str:
    m = re.search(r"GLOBAL__N_1\d+([a-z_0-9]+?_kernel)", n)
    if m:
        return m.group(1)
    if n.startswith("Cijk"):
        return "rocBLAS/Tensile GEMM"
    if "copyBuffer" in n:
        return "copyBuffer"
    m = re.search(r"at6native\d+([a-zA-Z_0-9]+?)(I|E)", n)
    if m:
        f = re.search(r"(FillFunctor|CUDAFunctor_add|BinaryFunctor|MulFunctor|launch_clamp|sum_|where|masked_fill|copy|cat|index|gather|scatter)", n)
        return "aten:" + m.group(1)[:28] + (":" + f.group(1) if f else "")
    return n[:40]


def main(path, n_top=25):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda stem: next(x for x in tabs if x.startswith(stem))
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    # a training step's stepping pass launches train_chain_fwd_kernel once per closed-loop step: replays = runs of those
    tc = [i for i, r in enumerate(rows) if "train_chain_fwd_kernel" in r[0]]
    if not tc:
        print("no train_chain_fwd_kernel launches: not a training trace")
        return
    # one train_chain_bwd_kernel per step; the optimizer's multi-tensor kernels (clip + AdamW, eager, between two replays) bound a step
    bw = [i for i, r in enumerate(rows) if "train_chain_bwd_kernel" in r[0]]
    is_opt = lambda n: "multi_tensor_apply" in n or "adam" in n.lower()
    b = bw[-1]
    end = next((i for i in range(b, len(rows)) if is_opt(rows[i][0])), len(rows))
    start = [j for j in tc if j < b and (len(bw) < 2 or j > bw[-2])][0]
    while start > 0 and not is_opt(rows[start - 1][0]) and (len(bw) < 2 or start - 1 > bw[-2]):
        start -= 1
    step = rows[start:end]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f"last step: {len(step)} launches, span {(t1 - t0) / 1e6:.2f} ms")

    def stats(seg, title):
        if not seg:
            return
        s0, s1 = seg[0][1], max(r[2] for r in seg)
        ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
        busy, depth, last, gaps = 0, 0, s0, []
        for tt, d in ev:
            if depth > 0:
                busy += tt - last
            elif tt > last:
                gaps.append((tt - last, tt))
            last = tt
            depth += d
        tot = sum(r[2] - r[1] for r in seg)
        print(f"\n== {title}: span {(s1 - s0) / 1e6:8.2f} ms, busy {busy / 1e6:8.2f}, idle {(s1 - s0 - busy) / 1e6:7.2f} in {len(gaps)} gaps, "
              f"summed kernel time {tot / 1e6:8.2f} (concurrency {tot / max(busy, 1):.2f}), {len(seg)} launches, mean {tot / len(seg) / 1e3:.1f} us")
        cls = collections.OrderedDict()
        for n, s, e in seg:
            c = cls.setdefault(short(n), [0, 0])
            c[0] += 1
            c[1] += e - s
        for k, (c, d) in sorted(cls.items(), key=lambda kv: -kv[1][1])[:n_top]:
            print(f"   {d / 1e6:8.2f} ms {c:6d} x {d / c / 1e3:8.1f} us  {100 * d / tot:5.1f} %  {k}")
        big = sorted(gaps, reverse=True)[:6]
        if big:
            after = {tt: next((short(r[0]) for r in seg if r[1] == tt), "?") for _, tt in big}
            print("   longest gaps (us, before kernel): " + ", ".join(f"{g / 1e3:.0f} -> {after[tt]}" for g, tt in big))
        hist = collections.Counter(min(int(g / 1e3) // 2 * 2, 20) for g, _ in gaps)
        print("   gap histogram (us bucket: count, total ms): " + ", ".join(
            f"{k}{'+' if k == 20 else ''}: {v} / {sum(g for g, _ in gaps if min(int(g / 1e3) // 2 * 2, 20) == k) / 1e6:.2f}" for k, v in sorted(hist.items())))

    stats(step, "whole step")
    tcs = [i for i, r in enumerate(step) if "train_chain_fwd_kernel" in r[0]]
    bwd = next((i for i, r in enumerate(step) if "train_chain_bwd_kernel" in r[0]), len(step))
    if len(tcs) >= 3:
        # the stepping pass: from the first per-step launch group to the last single-step train_chain_fwd (the batched pass launches it once more)
        first_step, last_step = tcs[0], tcs[-2]
        stats(step[:first_step], "before the stepping pass (map encoder, lights' pre-compute, posterior, navi predictor, first lights' piece)")
        stats(step[first_step:last_step + 1], "stepping pass (incl. what runs beside it on the side stream)")
        stats(step[last_step + 1:bwd], "batched differentiated forward + loss")
        stats(step[bwd:], "backward")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
