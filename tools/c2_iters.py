#!/usr/bin/env python3
"""C2 cold sweep: histogram of (nfev, iterations) and where along the sweep the third evaluation was a full pass
(iterations 4: the wavefront could not take the residual-only confirming pass)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, "cuda:0")
info = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain_len=1, predictor=False).info()
pairs, counts = np.unique(np.stack([info["nfev"], info["iterations"]], 1), axis=0, return_counts=True)
print("(nfev, iterations): count", {tuple(int(v) for v in p): int(c) for p, c in zip(pairs, counts)})
it = info["iterations"].reshape(-1, 16)
print("wavefronts with a problem at 4 iterations:", int((it.max(1) >= 4).sum()), "of", it.shape[0])
print("iterations by 1/16 of the sweep:", [round(float(info["iterations"][i:i + 1024].mean()), 2) for i in range(0, 16384, 1024)])
print("last_step percentiles:", np.percentile(info["last_step"], [0, 10, 50, 90, 100]))
