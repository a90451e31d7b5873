#!/usr/bin/env python3
"""Instruction counts per section of the lane kernel's full pass, from a kernel generated with OKX_DEV=lane_mark
(`OKX_DEV=lane_mark bash tools/lane_isa.sh dw` writes /tmp/q/lane_okx_lane_solve_u.s).  Sections are delimited by
`s_nop 11..16`: 1 rows (residuals, gradients, J^T r, diagonal), 2 LM decision, 3 factorisation + forward substitution,
4 backward substitution, 5 step bookkeeping, 6 rest of the kernel."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/q/lane_okx_lane_solve_u.s"
lines = [l for l in open(path) if re.match(r"^\s+[a-z]", l)]
names = {0: "before the pass", 1: "rows", 2: "LM decision", 3: "factor + forward", 4: "backward", 5: "step bookkeeping", 6: "after the pass"}
state = 0
count = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines:
    m = re.match(r"\s+s_nop (\d+)\s*$", l.split(";")[0] + "\n")
    if m and 11 <= int(m.group(1)) <= 16:
        state = int(m.group(1)) - 10
        continue
    op = l.split()[0]
    kind = ("fp64" if re.match(r"v_(fma|mul|add|fmac|max|min)_f64", op) else "agpr" if "accvgpr" in op
            else "lane" if re.match(r"v_(read|write)lane", op) else "scratch" if op.startswith("scratch") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith("global") else "salu" if op.startswith("s_") else "other")
    count[state] += 1
    kinds[state][kind] += 1
for k in sorted(count):
    print(f"{names[k]:20s} {count[k]:5d}  {dict(kinds[k])}")
