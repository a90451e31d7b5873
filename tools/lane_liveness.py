#!/usr/bin/env python3
"""Source-order liveness of the lane kernel's pass: number of live doubles per statement of the okx_lane_eval body in
/tmp/q/lane.hip (first definition -> last use of every named double).  What a register allocator would need if the
compiler kept the emitted order."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/q/lane.hip").read()
body = src[src.index("okx_lane_eval(QEvalArgs a)"):]
body = body[:body.index("extern \"C\" __global__", 10)]
stmts = [s.strip() for s in re.split(r";\s*", body) if s.strip()]
ident = re.compile(r"\b[A-Za-z_][A-Za-z0-9_]*\b")
first, last = {}, {}
for k, s in enumerate(stmts):
    for name in ident.findall(s):
        if name not in first:
            first[name] = k
        last[name] = k
names = [n for n in first if re.match(r"^(_[a-z]+\d+|[ACLEpry]\d+(_\d+)?|gn\d+|dinv\d+|nx\d+|L\d+_\d+(_lo|_hi)?|hs\d+_\d+|hd\d+|tv\d+|ss|mres_new|pmin|pmax)$", n)]
events = [0] * (len(stmts) + 1)
for n in names:
    w = 0.5 if n.endswith(("_lo", "_hi")) else 1.0
    if n.endswith(("_lo", "_hi")):
        continue  # parked halves live in AGPRs
    for k in range(first[n], last[n] + 1):
        events[k] += w
peak = max(range(len(stmts)), key=lambda k: events[k])
print("statements", len(stmts), "peak live doubles", events[peak], "at", peak, stmts[peak][:80])
step = max(1, len(stmts) // 40)
for k in range(0, len(stmts), step):
    print(f"{k:5d} {events[k]:6.1f}  {stmts[k][:70]}")
