#!/usr/bin/env python3
"""The streaming kernels either side of the solve at C5 scale (a million double-wishbone states), HIP-event time and
algorithmic GB/s of each: okx_expand_positions_batch (free coordinates -> records: the receiving side of the all-gather),
okx_tangent_batch, okx_corner_metrics_batch without / with derivative columns, okx_axle_metrics_batch (C3 states).
   python3 tools/stream_rates.py [n_states]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.input import load_geometry
from open_kinematics_amd.metrics import axle_roles, axle_state_metrics, corner_roles, corner_state_metrics
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, geometry_path

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


p, t = bump_sweep_problem(n)
dp = DeviceProgram(p, dev)
res = dp.solve(torch.as_tensor(t, device=dev), chain_len=-1)
pos = res.positions
T, n_out, n_free = p.n_targets, p.n_out, p.n_free
free = pos[:, dp.free_out_index].contiguous()
out = torch.empty_like(pos)
rows = {}


def row(name, ms, bytes_per_state, states=n):
    rows[name] = {"ms": round(ms, 4), "states_per_s": states / ms * 1e3, "bytes_per_state": bytes_per_state,
                  "algorithmic_gbs": round(bytes_per_state * states / ms / 1e6, 1), "hbm_frac": round(bytes_per_state * states / ms / 1e6 / 8000.0, 3)}
    print(f"{name:34s} {ms:8.3f} ms  {states / ms * 1e3:10.3g} states/s  {bytes_per_state:5d} B/state  {rows[name]['algorithmic_gbs']:7.0f} GB/s  {rows[name]['hbm_frac']:.3f} of 8 TB/s")


ms = timed(lambda: dp.expand(free, out=out))
print("expand vs the solver's records: max |diff|", float((out - pos).abs().max()), "(bit-identical within one kernel family; this solve ran the", dp.kernel, "family)")
row("expand (free -> records)", ms, 24 * n_free + 24 * n_out)
ms = timed(lambda: out.copy_(pos))
row("torch copy of the records (ref.)", ms, 48 * n_out)
tan, _ = dp.tangents(pos)
ms = timed(lambda: dp.tangents(pos))
row("tangents", ms, 24 * n_out * (1 + T) + 24)
roles = corner_roles(load_geometry(geometry_path("geometry.yaml")), p)
ms = timed(lambda: corner_state_metrics(roles, pos, None))
row("corner metrics", ms, 24 * n_out + 152)
ms = timed(lambda: corner_state_metrics(roles, pos, tan))
row("corner metrics + derivatives", ms, 24 * n_out * (1 + T) + 152 * (1 + T))
del tan, out, free, res, pos
torch.cuda.empty_cache()

side = int(round((n // 4) ** 0.5))
pa, ta = axle_grid_problem(side, side)
dpa = DeviceProgram(pa, dev)
resa = dpa.solve(torch.as_tensor(ta, device=dev), chain_len=-1)
axle = load_geometry(geometry_path("axle_geometry_rocker.yaml"))
left, right = axle_roles(axle, pa)
ms = timed(lambda: axle_state_metrics(left, right, resa.positions))
row("axle metrics (C3 states)", ms, 24 * pa.n_out + 56, side * side)
freea = resa.positions[:, dpa.free_out_index].contiguous()
outa = torch.empty_like(resa.positions)
ms = timed(lambda: dpa.expand(freea, out=outa))
print("axle expand vs records: max |diff|", float((outa - resa.positions).abs().max()))
row("expand, axle (C3 states)", ms, 24 * pa.n_free + 24 * pa.n_out, side * side)
print(json.dumps({"n_states": n, "rows": rows}))
