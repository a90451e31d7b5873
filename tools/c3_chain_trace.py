#!/usr/bin/env python3
"""Pass-by-pass LM record of the first problems of a C3 chain (chain_len = -1)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem
program, targets = axle_grid_problem(256, 256)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=4)
for prob in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]:
    tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), prob)
    dp.solve(t, chain_len=-1, predictor=False)
    torch.cuda.synchronize()
    a = tr.cpu().numpy()
    print(f"problem {prob}: pass: mode Ft Fc lambda step rho accept done")
    for k in range(0, 10):
        if np.any(a[k] != 0): print("  ", k, a[k])
    dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
