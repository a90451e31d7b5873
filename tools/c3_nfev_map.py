#!/usr/bin/env python3
"""Where the evaluations of the C3 grid's cold starts are: histogram, 16 x 16 tile means over heave x roll, and the
maximum per wave unit of eight (what the lockstep of a unit costs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem
program, targets = axle_grid_problem(256, 256)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
res = dp.solve(t, chain_len=1, predictor=False)
info = res.info()
nfev = info["nfev"].reshape(256, 256)
print("hist", {int(k): int((nfev == k).sum()) for k in np.unique(nfev)})
print("targets columns: min/max", targets.min(axis=0), targets.max(axis=0))
tg = targets.reshape(256, 256, -1)
for k in np.unique(nfev):
    if k >= 5:
        idx = np.argwhere(nfev == k)
        print(k, "rows(range)", idx[:, 0].min(), idx[:, 0].max(), "cols(range)", idx[:, 1].min(), idx[:, 1].max(), "n", len(idx))
# coarse map: mean nfev over 16x16 tiles
m = nfev.reshape(16, 16, 16, 16).mean(axis=(1, 3))
np.set_printoptions(precision=1, linewidth=200)
print(m)
# units of 8 consecutive problems: max nfev per unit
u = nfev.reshape(-1, 8).max(axis=1)
print("unit max hist", {int(k): int((u == k).sum()) for k in np.unique(u)})
print("flat index of a few nfev>=6:", np.flatnonzero(nfev.reshape(-1) >= 6)[:10])
print("targets there:", targets[np.flatnonzero(nfev.reshape(-1) >= 6)[:5]])
print("design targets", [float(np.dot(program.design_pos[p], d)) for p, d in zip(program.tgt_point, program.tgt_dir)])
