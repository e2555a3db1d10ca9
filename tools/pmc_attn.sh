#!/bin/bash
# SQ counters of the attention kernel at the WOSAC shape (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/pmc_attn; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $out/avail.txt
wc -l $out/avail.txt
cmd="bench.py --no-cpu-baseline --no-wosac-shape --no-lights-ahead --no-graph --profile-steps 0 --steps 4 --warmup 2 --agents 128 --rollouts 32 --scenes 1"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p -- python3 $cmd > $out/p$i.log 2>&1
  db=$(find $out/p$i -name '*.db' | head -1)
  python3 - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda stem: next(x for x in tabs if x.startswith(stem))
pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
q = (f"select s.kernel_name, p.name, count(*), avg(e.value), avg(d.end-d.start) from {pe} e join {pi} p on e.pmc_id = p.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where s.kernel_name like '%knarpe_attn_kernel%' group by s.kernel_name, p.name")
for r in db.execute(q): print(r[0][20:50], r[1], "n", r[2], "avg", round(r[3], 1), "dur_ns", round(r[4]))
PY
  rm -rf $out/p$i
done
