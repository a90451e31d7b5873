#!/usr/bin/env python3
"""Evaluations per chain position on the C3 axle grid (chain_len = -1: one chain per resident problem slot)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem
program, targets = axle_grid_problem(256, 256)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
for cl in (-1, 8, 16, 32):
    res = dp.solve(t, chain_len=cl, predictor=False)
    i = res.info()
    L = 8 if cl == -1 else cl
    nf = i["nfev"].reshape(-1, L)
    it = i["iterations"].reshape(-1, L)
    print(f"chain_len={cl}: nfev by position {np.round(nf.mean(0), 2)}  iterations {np.round(it.mean(0), 2)}  mean {nf.mean():.3f}")
print("targets of the first chain:", targets[:9])
