#!/usr/bin/env python3
"""Evaluation histogram and pass-by-pass LM record of a few problems of the C3 axle grid, cold starts."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem
program, targets = axle_grid_problem(64, 64)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=4)
res = dp.solve(t, chain_len=1, predictor=False)
info = res.info(); nf = info["nfev"]; it = info["iterations"]
print("nfev histogram:", {int(k): int(v) for k, v in zip(*np.unique(nf, return_counts=True))})
print("iterations histogram:", {int(k): int(v) for k, v in zip(*np.unique(it, return_counts=True))})
for prob in [int(a) for a in sys.argv[1:]] or [0, 63, 2080, 4095]:
    tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), prob)
    dp.solve(t, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    a = tr.cpu().numpy()
    print(f"problem {prob} targets {targets[prob] - targets[2080]}: pass: mode Ft Fc lambda step rho accept done")
    for k in range(0, 14):
        if np.any(a[k] != 0): print("  ", k, a[k])
    dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
