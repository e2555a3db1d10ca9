#!/usr/bin/env python
"""Outline of a kernel in a gfx950 .s file: the memory / MFMA / wait instructions in program order, runs compressed.
    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only x.hip -o /tmp/x.s && python tools/isa_outline.py /tmp/x.s <symbol substring>"""
import re
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index(":", i)
j = s.index(".Lfunc_end", i)
body = s[i:j].splitlines()
print(len(body), "lines")
pat = re.compile(r"(global_load|buffer_load|s_waitcnt|v_mfma|ds_read_b64_tr|ds_write|ds_read|s_cbranch|\.LBB|v_sin|v_cos|s_barrier|global_store|scratch_|s_load)")
out, prev, cnt, start, first = [], None, 0, 0, ""
for n, l in enumerate(body):
    t = l.strip().split(";")[0].strip()
    if not pat.match(t):
        continue
    k = t.split()[0]
    if k.startswith("s_waitcnt") or k.startswith(".LBB") or k.startswith("s_cbranch"):
        k = t  # (keep every distinct wait / label / branch)
    if k == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{start:5d} {prev} x{cnt}")
        prev, cnt, start, first = k, 1, n, t
out.append(f"{start:5d} {prev} x{cnt}")
print("\n".join(out))
