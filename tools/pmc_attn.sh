#!/bin/bash
# SQ / cache counters of the attention kernel at the WOSAC shape (separate --pmc passes, kernel-trace only), reduced to
# gpurun_out/${TAG}_attn_counters.json (raw per-launch sums + VALU-busy, L2 hit rate, L2 request bytes). TAG=r02 by default.
# KPAT = SQL LIKE pattern of the kernel (default: the wave-per-row VALU kernel), MINGRID = smallest grid (threads) counted,
# EXTRA = further bench.py arguments (e.g. "--kv-bf16 --attn-mfma 1" with KPAT=%knarpe_attn_mfma_kernel% MINGRID=65536).
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${TAG:-r02}
out=gpurun_out/pmc_attn; mkdir -p $out
kpat=${KPAT:-%knarpe_attn_kernelILi1E%}
mingrid=${MINGRID:-262144}
cmd="bench.py --no-cpu-baseline --no-wosac-shape --no-lights-ahead --no-graph --profile-steps 0 --steps 4 --warmup 2 --agents 128 --rollouts 32 --scenes 1 --new-scenes 0 --detail-file - ${EXTRA}"
i=0
: > $out/rows.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p -- python3 $cmd > $out/p$i.log 2>&1
  db=$(find $out/p$i -name '*.db' | head -1)
  python3 - "$db" "$kpat" "$mingrid" >> $out/rows.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda stem: next(x for x in tabs if x.startswith(stem))
pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
q = (f"select s.kernel_name, p.name, count(*), sum(e.value), count(distinct d.id), avg(d.end-d.start) from {pe} e join {pi} p on e.pmc_id = p.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where s.kernel_name like '{sys.argv[2]}' "
     f"and d.grid_size_x >= {int(sys.argv[3])} group by p.name")
for r in db.execute(q): print(r[1], r[2], r[3], r[4], round(r[5]))
PY
  rm -rf $out/p$i
done
python3 - $out/rows.txt "$cmd" "$kpat" > gpurun_out/${tag}_attn_counters.json <<'PY'
import json, sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
c, dur = {}, {}
for name, n_samples, total, n_launch, d in rows:
    c[name] = float(total) / max(int(n_launch), 1)  # per launch, summed over the samples (shader engines x XCDs)
    dur[name] = int(d)
out = {"collected_with": "rocprofv3 --pmc <set> --kernel-trace (one pass per counter set, tools/pmc_attn.sh) -- python3 " + sys.argv[2],
       "kernel": sys.argv[3] + " (wave per row), agents' launches at the WOSAC shape: 4096 source rows, 4 self launches (25 pairs per row) + 4 cross launches (89)",
       "per_launch": c, "launch_ns_under_pmc": dur}
simd = 1024.0
if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
    # rocprofiler's VALUBusy: 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE (GRBM_GUI_ACTIVE per sample: divide the sum by its sample count)
    ns = next(int(r[1]) / max(int(r[3]), 1) for r in rows if r[0] == "GRBM_GUI_ACTIVE")
    gui = c["GRBM_GUI_ACTIVE"] / ns
    out["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd / gui
    out["gui_active_cycles"] = gui
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    out["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
if "TCP_TCC_READ_REQ_sum" in c:
    out["l2_read_requests"] = c["TCP_TCC_READ_REQ_sum"]
if "SQ_INSTS_VALU" in c:
    # the agents' launches of a step: 4 self launches (25 pairs per row) + 4 cross launches (64 map + 25 light pairs): 57 pairs per row on average
    out["valu_insts_per_pair"] = c["SQ_INSTS_VALU"] * 64.0 / (4096 * 57.0) / 64.0
    out["valu_insts_per_wave_pass"] = c["SQ_INSTS_VALU"] / 4096.0 / (57.0 / 8.0)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "gui_active_cycles" in out:
    out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd / out["gui_active_cycles"]  # (per-SIMD busy cycles over the launch's cycles)
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    out["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
    out["active_inst_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
out["kv_bf16"] = "--kv-bf16" in sys.argv[2]
print(json.dumps(out, indent=1))
PY
cat gpurun_out/${tag}_attn_counters.json | head -50
