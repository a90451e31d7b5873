#!/usr/bin/env python3
"""LM pass record of a few problems of the C2 cold sweep (okx_debug_quad_trace) and the nfev histogram."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=4)
res = dp.solve(t, chain_len=1, predictor=False)
nf = res.info()["nfev"]
print("nfev histogram:", {int(k): int(v) for k, v in zip(*np.unique(nf, return_counts=True))})
print("nfev by eighth of the sweep:", [round(float(nf[i:i + 2048].mean()), 2) for i in range(0, 16384, 2048)])
for prob in [int(a) for a in sys.argv[1:]] or [100, 4000, 7000, 16000]:
    tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), prob)
    dp.solve(t, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    a = tr.cpu().numpy()
    print(f"problem {prob} (bump {targets[prob,1]-targets[7022,1]:+.1f} mm): pass: mode Ft Fc lambda step rho accept done")
    for k in range(0, 12):
        if np.any(a[k] != 0): print("  ", k, a[k])
    dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
