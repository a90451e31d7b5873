#!/usr/bin/env python3
"""
Stress run (not a test): perturbed double-wishbone geometries x random 2-D target boxes, chained and independent,
with and without the model; counts non-accepted solves and compares a sample with the oracle's MINPACK.
  tools/stress.py [n_geometries] [steps]
"""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import ensemble_problem
from oracle.oracle import Oracle

n_geo = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(11)
program, table, rel = ensemble_problem(n_geo, steps, sigma=2.0, seed=5)
dp = DeviceProgram(program, "cuda:0")
gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda:0"))
base = torch.stack([gpos[:, program.tgt_point[k]] @ torch.as_tensor(program.tgt_dir[k], device="cuda:0") for k in range(program.n_targets)], 1)
# random box per geometry: rack +-25 mm, bump -75..+95 mm, traversed along a random straight line
lo = np.stack([rng.uniform(-25, 0, n_geo), rng.uniform(-75, -20, n_geo)], 1)
hi = np.stack([rng.uniform(0, 25, n_geo), rng.uniform(20, 95, n_geo)], 1)
s = np.linspace(0.0, 1.0, steps)[None, :, None]
rel_t = lo[:, None, :] + s * (hi - lo)[:, None, :]
targets = (base[:, None, :] + torch.as_tensor(rel_t, device="cuda:0")).reshape(-1, program.n_targets).contiguous()
ref = None
for cl in (1, -1, steps):
    t0 = time.perf_counter()
    res = dp.solve(targets, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=steps, chain_len=cl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    info = res.info(); ok = res.accepted(info)
    if ref is None: ref = res.positions.clone()
    print(f"chain_len={cl:4d}: {targets.shape[0]} solves, not accepted {int((~ok).sum())}, evals {info['nfev'].mean():.2f} (max {info['nfev'].max()}), "
          f"max residual {info['max_residual'].max():.2e}, vs independent {float((res.positions - ref).abs().max()):.1e}, {dt*1e3:.1f} ms")
pos = ref.cpu().numpy().reshape(n_geo, steps, program.n_out, 3); th = targets.cpu().numpy().reshape(n_geo, steps, -1)
worst = 0.0
for g in rng.choice(n_geo, 12, replace=False):
    gp, rp = Oracle(program).rebind(table[g])
    pick = np.linspace(0, steps - 1, 6).astype(int)
    orc = Oracle(dataclasses.replace(program, design_pos=gp, row_param=rp)).sweep(th[g][pick], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert orc.first_failed_step == -1
    worst = max(worst, float(np.max(np.abs(pos[g][pick] - orc.positions))))
print(f"oracle sample (12 geometries x 6 steps): max |device - oracle| = {worst:.2e} mm")

# ---- composed axles in pair mode: perturbed geometries x random heave / roll lines, against the interpreter ----
#      rocker + U-bar axle (one joining row), T-bar axle (three), T-bar + heave-link axle (three, 11 free points per half)
from open_kinematics_amd.workloads import axle_grid_problem

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden

axles = [("rocker + U-bar axle", axle_grid_problem(2, 2)[0])]
for fixture in ("t_axle_t_bar_roll", "t_axle_t_bar_heave"):
    axles.append((fixture, load_golden(fixture)[1].with_line_mode("pinned")))
for label, program in axles:
    dp = DeviceProgram(program, "cuda:0")
    n_axle, s_axle = max(n_geo // 8, 16), 64
    hard = np.repeat(program.design_pos[None], n_axle, axis=0)
    moving = np.array([i for i in range(program.n_points) if program.role[i] != 2])
    hard[1:, moving] += rng.normal(0.0, 0.75, (n_axle - 1, len(moving), 3))
    gpos, gparam = dp.rebind(torch.as_tensor(hard, device="cuda:0"))
    base = torch.stack([gpos[:, program.tgt_point[k]] @ torch.as_tensor(program.tgt_dir[k], device="cuda:0") for k in range(program.n_targets)], 1)
    heave0, heave1 = rng.uniform(-35, 0, n_axle), rng.uniform(0, 35, n_axle)
    roll0, roll1 = rng.uniform(-18, 0, n_axle), rng.uniform(0, 18, n_axle)
    u = np.linspace(0.0, 1.0, s_axle)[None, :]
    heave = heave0[:, None] + u * (heave1 - heave0)[:, None]
    roll = roll0[:, None] + u * (roll1 - roll0)[:, None]
    rel = np.zeros((n_axle, s_axle, program.n_targets))
    rel[:, :, 0], rel[:, :, 1] = heave + roll, heave - roll   # left / right wheel-centre z; rack held
    targets = (base[:, None, :] + torch.as_tensor(rel, device="cuda:0")).reshape(-1, program.n_targets).contiguous()
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s_axle)
    wave = dp.solve(targets, kernel="single", **kw)
    print(f"{label} ({dp.kernel}, n = {program.n_vars}): interpreter not accepted {int((~wave.accepted(wave.info())).sum())} of {targets.shape[0]}")
    for cl in (1, -1, s_axle):
        res = dp.solve(targets, chain_len=cl, **kw)
        info = res.info(); ok = res.accepted(info)
        both = torch.as_tensor(ok & wave.accepted(wave.info()), device="cuda:0")
        print(f"  pair mode chain_len={cl:3d}: not accepted {int((~ok).sum())}, evals {info['nfev'].mean():.2f} (max {info['nfev'].max()}), "
              f"max |quad - interpreter| = {float((res.positions - wave.positions)[both].abs().max()):.1e}")
