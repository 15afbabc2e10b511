#!/usr/bin/env python3
"""Compact instruction-class trace of one kernel from a hipcc -S listing: M = MFMA, v = plain vector, e = transcendental, d = LDS,
g = global/buffer memory, s = scalar, w = s_waitcnt, B = barrier, | = label / branch.  Shows at a glance whether vector work is
interleaved with the matrix instructions (same-wave interleaving is what hides it: tools/ubench/mfma_valu_interleave.hip).
    python tools/isa_trace.py /tmp/attn_split.s k_attention_split32pILi8E"""
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(_Z\w*" + re.escape(name) + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M)
body = m.group(2)
out = []
for line in body.splitlines():
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            out.append("\n" + t + " ")
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): c = "e"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("ds_"): c = "d"
    elif op.startswith(("global_", "buffer_", "flat_")): c = "g"
    elif op.startswith("s_waitcnt"): c = "w"
    elif op.startswith("s_barrier"): c = "B"
    elif op.startswith(("s_cbranch", "s_branch")): c = "|"
    elif op.startswith("s_"): c = "s"
    else: c = "?"
    out.append(c)
print("".join(out))
