#!/usr/bin/env python3
"""Compare the quad kernel's r / J^T J / J^T r / LDL^T step with the generic kernel's at random points."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem, macpherson_grid_problem

which = sys.argv[1] if len(sys.argv) > 1 else "dw"
mode = sys.argv[2] if len(sys.argv) > 2 else "pinned"
program, targets = bump_sweep_problem(40, line_mode=mode) if which == "dw" else macpherson_grid_problem(6, 6, line_mode=mode)
dp = DeviceProgram(program, "cuda:0")
print("kernel", dp.kernel, dp.kernel_note)
rng = np.random.default_rng(0)
b = 36
x0 = program.design_pos[program.free_point].reshape(-1)
x = x0[None] + rng.normal(0, 5.0, (b, program.n_vars))
t = targets[:b]
r_w, ata_w, atr_w = dp.normal_equations(x, t)
lam = 1e-3
r_q, ata_q, atr_q, dx_q = dp.quad_eval(x, t, lam)
r_w, ata_w, atr_w, r_q, ata_q, atr_q, dx_q = [v.cpu().numpy() for v in (r_w, ata_w, atr_w, r_q, ata_q, atr_q, dx_q)]
np.set_printoptions(linewidth=200, precision=4)
print("max |r_q - r_w| per row:", np.abs(r_q - r_w).max(axis=0))
print("max |atr| diff per var:", np.abs(atr_q - atr_w).max(axis=0))
d = np.abs(ata_q - ata_w).max(axis=0)
print("ata diff (block max):")
n = program.n_vars
print(d.reshape(n // 3, 3, n // 3, 3).max(axis=(1, 3)))
dx_ref = np.stack([-np.linalg.solve(ata_w[k] + lam * np.eye(n), atr_w[k]) for k in range(b)])
print("max |dx_q - dx_ref| per var:", np.abs(dx_q - dx_ref).max(axis=0))
print("dx scale", np.abs(dx_ref).max())
