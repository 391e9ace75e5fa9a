"""Instruction mix per basic block of one kernel in a device assembly listing
(hipcc --cuda-device-only -S):  python tools/asm_mix.py file.s <substring of the kernel symbol>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") or (l.startswith("_Z") and key in l and ": " in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = [], Counter(), "entry"
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"):
            pass
        else:
            continue
    if l.endswith(":") or re.match(r"^\.LBB\S+:", l):
        if sum(cur.values()):
            blocks.append((name, cur))
        cur, name = Counter(), l.split(":")[0]
        continue
    op = l.split()[0]
    kind = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else
            "wait" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else "other")
    cur[kind] += 1
    if op.startswith("v_pk_"):
        cur["pk"] += 1
    if op.startswith("v_mov"):
        cur["mov"] += 1
    if op.startswith(("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_fmac", "v_pk_")):
        cur["fp"] += 1
if sum(cur.values()):
    blocks.append((name, cur))
tot = Counter()
for name, c in blocks:
    tot.update(c)
    if sum(c.values()) >= 40:
        print("%-12s %s" % (name, dict(c)))
print("total       ", dict(tot))
