#!/usr/bin/env python3
"""Chain-head predictor: same solutions, fewer passes.  tools/predictor_check.py [dw|mac|axle]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

cases = {"dw": lambda: bump_sweep_problem(16384), "mac": lambda: macpherson_grid_problem(128, 128),
         "axle": lambda: axle_grid_problem(90, 90)}
for name in (sys.argv[1:] or list(cases)):
    program, targets = cases[name]()
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    base = dp.solve(t, chain_len=1, predictor=False)
    pred = dp.solve(t, chain_len=1, predictor=True)
    torch.cuda.synchronize()
    bi, pi = base.info(), pred.info()
    diff = float((base.positions - pred.positions).abs().max())
    print(f"{name}: B={t.shape[0]} kernel={dp.kernel}  max |x_pred - x_cold| = {diff:.2e}  nfev cold {bi['nfev'].mean():.2f} -> predictor {pi['nfev'].mean():.2f}"
          f"  accepted {bool(base.accepted(bi).all())}/{bool(pred.accepted(pi).all())}  max residual {pi['max_residual'].max():.1e}")
    out = torch.empty_like(base.positions); info = torch.empty_like(base.info_raw)
    for cl in (1, -1):
        a = timed(dp.plan(t, out=out, info_out=info, chain_len=cl, predictor=False))
        b = timed(dp.plan(t, out=out, info_out=info, chain_len=cl, predictor=True))
        print(f"    chain_len={cl:2d}: {a:8.2f} us -> {b:8.2f} us   ({t.shape[0]/a*1e6:.3e} -> {t.shape[0]/b*1e6:.3e} solves/s)")
