#!/usr/bin/env python3
"""The quad kernel's cold body (fast loop + redo path) against its general body on independent solves: C2, a MacPherson
bump x rack grid out to the edge of the rack's reach (rejected steps: the redo path), a perturbed ensemble (per-geometry
tables), ragged batch sizes.  Same answers (<= 1e-10 mm), same flags; time per launch of both.
   python3 tools/cold_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram

dev = torch.device("cuda:0")
INFO = [("max_residual", "f8"), ("cost", "f8"), ("last_step", "f8"), ("iterations", "i4"), ("nfev", "i4"), ("flags", "i4"), ("reserved", "i4")]


def run(dp, t, n_out, timed, **kw):
    n = t.shape[0]
    out = torch.empty((n, n_out, 3), dtype=torch.float64, device=dev)
    info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
    launch = dp.plan(t, out=out, info_out=info, chain_len=1, predictor=False, kernel="quad", **kw)
    launch()
    torch.cuda.synchronize()
    ms = float("nan")
    if timed:
        for _ in range(200):
            launch()
        _, ms = bench.time_launches(launch, 1000, 50, dev)
    torch.cuda.synchronize()
    return out.cpu().numpy().copy(), np.frombuffer(info.cpu().numpy().tobytes(), dtype=INFO).copy(), ms


def case(label, program, targets, timed=True, **kw):
    dp = DeviceProgram(program, dev)
    t = torch.as_tensor(targets, device=dev)
    geo = {k: torch.as_tensor(v, device=dev) for k, v in kw.items() if isinstance(v, np.ndarray)}
    other = {k: v for k, v in kw.items() if not isinstance(v, np.ndarray)}
    os.environ.pop("OKX_DEV", None)
    pc, ic, mc = run(dp, t, program.n_out, timed, **geo, **other)
    os.environ["OKX_DEV"] = "no_cold"
    pg, ig, mg = run(dp, t, program.n_out, timed, **geo, **other)
    os.environ.pop("OKX_DEV", None)
    ok = (ig["flags"] & 7) == 1
    d = np.abs(pc - pg).reshape(len(ok), -1).max(axis=1)
    print(f"{label:46s} n={len(ok):7d} cold {1e3 * mc:8.2f} us  general {1e3 * mg:8.2f} us  nfev {ic['nfev'].mean():.3f} / {ig['nfev'].mean():.3f}"
          f"  converged {ok.mean():.4f}  max|d| (converged) {d[ok].max() if ok.any() else 0:.2e}  flags equal {np.array_equal(ic['flags'], ig['flags'])}"
          f"  nfev differs on {int((ic['nfev'] != ig['nfev']).sum())}")
    dp.close()
    return d[ok].max() if ok.any() else 0.0, np.array_equal(ic["flags"] & 7, ig["flags"] & 7)


worst, same = 0.0, True
p, t = W.bump_sweep_problem(16384)
r = case("C2 double wishbone 16384-step bump sweep", p, t); worst = max(worst, r[0]); same &= r[1]
for n in (1, 15, 17, 1000):
    p, t = W.bump_sweep_problem(n)
    r = case(f"  the same, {n} steps", p, t, timed=False); worst = max(worst, r[0]); same &= r[1]
p, t = W.bump_sweep_problem(16384, line_mode="softnorm")
r = case("C2 with the reference's softnorm line row", p, t); worst = max(worst, r[0]); same &= r[1]
p, t = W.macpherson_grid_problem(128, 128)
r = case("C4 MacPherson 128 x 128 grid", p, t); worst = max(worst, r[0]); same &= r[1]
# out to (and beyond) the reach: bump +-120 mm, rack +-70 mm - rejected steps, failures
bump = np.linspace(-120.0, 140.0, 128)
rack = np.linspace(-75.0, 75.0, 128)
t2 = t.copy().reshape(128, 128, -1)
base = t.reshape(128, 128, -1)
mid_b, mid_r = base[:, 0, :].copy(), base[0, :, :].copy()
# targets are absolute: rebuild from the design values (centre of the grid) and new offsets
design = 0.5 * (base[0, 0] + base[-1, -1])
for i in range(128):
    for j in range(128):
        t2[i, j] = design
which_bump = int(np.argmax(np.abs(base[-1, 0] - base[0, 0])))
which_rack = 1 - which_bump
t2[:, :, which_bump] += bump[:, None]
t2[:, :, which_rack] += rack[None, :]
r = case("MacPherson grid beyond the reach (redo path)", p, t2.reshape(-1, t.shape[1])); worst = max(worst, r[0]); same &= r[1]
p, t = W.axle_grid_problem(128, 128)
r = case("C3 rocker + U-bar axle 128 x 128 grid (pair mode)", p, t); worst = max(worst, r[0]); same &= r[1]
for n in (1, 7, 9):
    r = case(f"  the same, {n} problems", p, t[:n], timed=False); worst = max(worst, r[0]); same &= r[1]
p, t = W.axle_grid_problem(256, 256)
r = case("C3 at full size, 256 x 256", p, t); worst = max(worst, r[0]); same &= r[1]
hr, rr = np.meshgrid(np.linspace(-75.0, 75.0, 64), np.linspace(-45.0, 45.0, 64), indexing="ij")  # heave x roll beyond the reach
base3 = 0.5 * (t[0] + t[-1])
t3 = np.stack([base3[0] + (hr + rr).ravel(), base3[1] + (hr - rr).ravel(), np.full(hr.size, base3[2])], axis=1)
r = case("axle grid beyond the reach (redo path, pair mode)", p, t3); worst = max(worst, r[0]); same &= r[1]
program, table, rel = W.ensemble_problem(256, 64)
dpe = DeviceProgram(program, dev)
gpos, gparam = dpe.rebind(torch.as_tensor(table, device=dev))
base_t = torch.stack([gpos[:, program.tgt_point[k]] @ torch.as_tensor(program.tgt_dir[k], device=dev) for k in range(program.n_targets)], 1)
te = (base_t[:, None, :] + torch.as_tensor(rel, device=dev)[None]).reshape(-1, program.n_targets).contiguous().cpu().numpy()
r = case("ensemble 256 geometries x 64 steps", program, te, geom_pos=gpos.cpu().numpy(), geom_row_param=gparam.cpu().numpy(), steps_per_geometry=64)
worst = max(worst, r[0]); same &= r[1]
dpe.close()
print(f"worst difference on converged problems {worst:.2e} mm; outcome flags equal everywhere: {same}")
sys.exit(0 if worst <= 1e-10 and same else 1)
