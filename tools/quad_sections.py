#!/usr/bin/env python3
"""Instruction counts per section of the quad kernel's full pass, from a kernel generated with OKX_DEV=quad_mark
(`OKX_DEV=quad_mark,quad_no_light bash tools/quad_isa.sh dw` writes /tmp/q/u.s).  Sections are delimited by
`s_nop 11..17`: 1 row residual + gradient, 2 chain blocks + J^T r, 3 J^T J, 4 after the row, 5 factorisation,
6 substitution, 7 after the solve."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/q/u.s"
lines = [l for l in open(path) if re.match(r"^\s+[a-z]", l)]
names = {0: "before the first row", 1: "row residual + gradient", 2: "chain blocks + J^T r", 3: "J^T J", 4: "after the row",
         5: "factorisation", 6: "substitution", 7: "after the solve"}
state, last4 = 0, False
count = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines:
    m = re.match(r"\s+s_nop (\d+)\s", l)
    if m and 11 <= int(m.group(1)) <= 17:
        state = int(m.group(1)) - 10
        continue
    op = l.split()[0]
    kind = ("fp64" if re.match(r"v_(fma|mul|add|fmac)_f64", op) else "dpp" if "dpp" in op else "agpr" if "accvgpr" in op
            else "nop" if op == "s_nop" else "other")
    count[state] += 1
    kinds[state][kind] += 1
for k in sorted(count):
    print(f"{names[k]:28s} {count[k]:5d}  {dict(kinds[k])}")
