#!/usr/bin/env python3
"""Per-pass Levenberg-Marquardt record of single problems of the C3 axle grid (okx_debug_quad_trace; the general body, which
the cold body matches bit for bit): mode, trial cost, accepted cost, damping, step, gain ratio, accepted, done.
    python tools/c3_lm_trace.py [problem ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem

program, targets = axle_grid_problem(256, 256)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=4)
for problem in [int(a) for a in sys.argv[1:]] or [0, 128 * 256 + 128, 200 * 256 + 30, 65535]:
    tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), problem)
    res = dp.solve(t, chain_len=1, kernel="quad")
    torch.cuda.synchronize()
    a = tr.cpu().numpy()
    info = res.info()[problem]
    print(f"problem {problem} targets-rel {targets[problem] - targets[128 * 256 + 128]} nfev {info['nfev']} iterations {info['iterations']} last_step {info['last_step']:.2e}")
    print("  pass mode Ft Fc lambda step rho accept done")
    for k in range(1, 12):
        if a[k].any():
            print("  ", k, a[k])
dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
