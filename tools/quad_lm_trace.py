#!/usr/bin/env python3
"""Per-pass Levenberg-Marquardt record of ONE problem in the quad kernel (okx_debug_quad_trace):
mode, trial cost, accepted cost, damping, step, gain ratio, accepted, done.  Usage:
    python tools/quad_lm_trace.py [problem_index]    (reference softnorm line rows, 101-step C1 sweep)"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(101, line_mode="softnorm")
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), int(sys.argv[1]) if len(sys.argv) > 1 else 75)
res = dp.solve(t, chain_len=1, kernel="quad", max_iter=40, step_tol=1e-8)
torch.cuda.synchronize()
np.set_printoptions(linewidth=200, precision=6)
a = tr.cpu().numpy()
print("pass mode Ft Fc lambda step rho accept done")
for k in range(1, 60):
    print(k, a[k])
