#!/usr/bin/env python3
"""developer aid: static census of the loops of one kernel in a built object (backward branches -> [target, branch] bodies): size and
instruction mix per loop, innermost first.  usage: loop_census.py <object-or-so> <kernel-name-substring> [min_size]"""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_spill_exec as c
from collections import Counter
txt = c.disassemble(sys.argv[1]); name = sys.argv[2]; min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = txt.split("\n")
s = [i for i, l in enumerate(lines) if name in l and l.rstrip().endswith(">:")][0]
e = next((i for i in range(s + 1, len(lines)) if lines[i].rstrip().endswith(">:")), len(lines))
ins = []
for l in lines[s + 1:e]:
    m = re.match(r"\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr2i = {a: i for i, (a, _, _) in enumerate(ins)}
def klass(op):
    if "mfma" in op: return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"
loops = []
for i, (a, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", lines[s + 1 + i]) if False else None
for i, l in enumerate(lines[s + 1:e]):
    m = re.match(r"\s*(s_cbranch\S*|s_branch)\s+(\d+)\s*//\s*([0-9A-Fa-f]+):.*<[^>]*\+0x([0-9A-Fa-f]+)>", l)
    if not m: continue
    src = int(m.group(3), 16); base = ins[0][0]; tgt = base + int(m.group(4), 16)
    if tgt <= src and tgt in addr2i and src in addr2i: loops.append((addr2i[tgt], addr2i[src]))
loops.sort(key=lambda p: p[1] - p[0])
print(f"{name}: {len(ins)} instructions, {len(loops)} loops")
for lo, hi in loops:
    n = hi - lo + 1
    if n < min_size: continue
    inner = [(a, b) for a, b in loops if lo <= a and b <= hi and (a, b) != (lo, hi)]
    cnt = Counter(klass(op) for _, op, _ in ins[lo:hi + 1])
    f64 = sum(1 for _, op, _ in ins[lo:hi + 1] if op.endswith("_f64") and op.startswith("v_"))
    print(f"  [{ins[lo][0]:#x}..{ins[hi][0]:#x}] n={n:5d} inner={len(inner):2d} " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())) + (f" (f64 valu {f64})" if f64 else ""))
