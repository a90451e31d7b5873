import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16)
dp = DeviceProgram(program, "cuda:0")
n = program.n_vars
x0 = np.repeat(program.design_pos[program.free_point].reshape(1, -1), 16, axis=0)
r, ata, atr, dx = [v.cpu().numpy() for v in dp.quad_eval(x0, targets, 0.0)]
dmax = np.max(np.diagonal(ata, axis1=1, axis2=2), axis=1)
print("dmax", dmax[:3], "cost0", 0.5 * (r ** 2).sum(1)[:3])
lam = 1e-6 * dmax[0]
r, ata, atr, dx = [v.cpu().numpy() for v in dp.quad_eval(x0, targets, lam)]
dx_ref = np.stack([-np.linalg.solve(ata[k] + lam * np.eye(n), atr[k]) for k in range(16)])
print("max|dx - ref|", np.abs(dx - dx_ref).max(), "step", np.abs(dx).max(1)[[0, 7, 15]])
r1 = dp.quad_eval(x0 + dx, targets, lam)[0].cpu().numpy()
r1w = dp.eval(x0 + dx, targets, jac=False)[0].cpu().numpy()
print("cost at x0+dx: quad", 0.5 * (r1 ** 2).sum(1)[[0, 7, 15]], "wave", 0.5 * (r1w ** 2).sum(1)[[0, 7, 15]])
r1 = dp.quad_eval(x0 + dx_ref, targets, lam)[0].cpu().numpy()
print("cost at x0+dx_ref: quad", 0.5 * (r1 ** 2).sum(1)[[0, 7, 15]])
print("cond", np.linalg.cond(ata[0] + lam * np.eye(n)))
np.set_printoptions(linewidth=200, precision=5)
print(dx[0]); print(dx_ref[0])
