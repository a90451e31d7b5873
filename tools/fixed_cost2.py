#!/usr/bin/env python3
"""The fixed part of a C2 launch, taken apart: a trivial kernel back to back (launch overhead), the solve cut short on the
shared first step with and without its record stores, 16 wavefronts against 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
dev = torch.device("cuda:0")
p, t = W.bump_sweep_problem(16384)
dp = DeviceProgram(p, dev)
tg = torch.as_tensor(t, device=dev)
out = torch.empty((tg.shape[0], p.n_out, 3), dtype=torch.float64, device=dev)
free = torch.empty((tg.shape[0], p.n_free, 3), dtype=torch.float64, device=dev)
info = torch.empty((tg.shape[0], 40), dtype=torch.uint8, device=dev)
one = torch.zeros(64, device=dev)
for _ in range(2000): one.add_(1.0)
wall, ms = bench.time_launches(lambda: one.add_(1.0), 2000, 100, dev)
print(f"{'trivial torch kernel (64 elements)':44s}: {1e3 * ms:6.2f} us per launch")
for label, n, kw in (("first step only, records", 16384, dict(out=out)), ("first step only, free coordinates", 16384, dict(out=free, output="free")),
                     ("first step only, no position stores", 16384, dict(output="none")),
                     ("first step only, records, 256 problems", 256, dict(out=out[:256])), ("first step only, no stores, 256 problems", 256, dict(output="none")),
                     ("full solve, records", 16384, dict(out=out, full=True)), ("full solve, no position stores", 16384, dict(output="none", full=True))):
    full = kw.pop("full", False)
    o = kw.pop("out", None)
    extra = {} if full else {"step_tol": 1e9}
    launch = dp.plan(tg[:n], out=o, info_out=info[:n], chain_len=1, predictor=False, **kw, **extra)
    for _ in range(500): launch()
    wall, ms = bench.time_launches(launch, 2000, 100, dev)
    print(f"{label:44s}: {1e3 * ms:6.2f} us per launch")
