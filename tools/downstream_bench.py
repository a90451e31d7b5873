#!/usr/bin/env python3
"""Throughput of the callers after the solve at C5 scale: tangents, corner metrics with derivative columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.input import load_geometry
from open_kinematics_amd.metrics import corner_roles, corner_state_metrics
from open_kinematics_amd.workloads import bump_sweep_problem, geometry_path
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
p, t = bump_sweep_problem(n)
dp = DeviceProgram(p, "cuda:0")
res = dp.solve(torch.as_tensor(t, device="cuda:0"), chain_len=-1)
roles = corner_roles(load_geometry(geometry_path("geometry.yaml")), p)
def timed(fn, reps=10):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, out
tt, (tan, _) = timed(lambda: dp.tangents(res.positions))
tm, _ = timed(lambda: corner_state_metrics(roles, res.positions, tan))
tm0, _ = timed(lambda: corner_state_metrics(roles, res.positions, None))
bt = 24 * p.n_out * (1 + p.n_targets) + 24
bm = 24 * p.n_out * (1 + p.n_targets) + 152 * (1 + p.n_targets)
print(f"{n} states: tangents {n/tt:.3g}/s ({tt*1e3:.3f} ms, {bt*n/tt/1e9:.0f} GB/s algorithmic); metrics+derivatives {n/tm:.3g}/s ({tm*1e3:.3f} ms, {bm*n/tm/1e9:.0f} GB/s); metrics only {n/tm0:.3g}/s ({tm0*1e3:.3f} ms, {(360+152)*n/tm0/1e9:.0f} GB/s)")
