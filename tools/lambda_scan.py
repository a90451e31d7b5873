#!/usr/bin/env python3
"""Cold-start time and evaluations against the initial damping lambda0 (okx_solve_opts.lambda0, relative to the largest diagonal
entry of J^T J at the design state) on C2 / C3 / C4, and the distance from the tightly converged point."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem
dev = "cuda:0"
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for name, make in (("C2", lambda: bump_sweep_problem(16384)), ("C3", lambda: axle_grid_problem(256, 256)), ("C4", lambda: macpherson_grid_problem(512, 512))):
    p, t = make()
    dp = DeviceProgram(p, dev)
    tt = torch.as_tensor(t, device=dev)
    tight = dp.solve(tt, chain_len=1, predictor=False, step_tol=1e-13, confirm_full_pass=True, max_iter=200).positions.clone()
    for lam in (None, 1e-2, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-10):
        kw = {} if lam is None else {"lambda0": lam}
        launch = dp.plan(tt, chain_len=1, predictor=False, **kw)
        ms = timed(launch)
        res = launch(); torch.cuda.synchronize()
        inf = res.info()
        print(f"{name} lambda0 {'default' if lam is None else f'{lam:7.0e}'}: {ms*1e3:8.2f} us, nfev {inf['nfev'].mean():.3f}, max |x - tight| {float((res.positions - tight).abs().max()):.2e}, converged {float(np.mean((inf['flags'] & 7) == 1)):.5f}", flush=True)
    dp.close()
