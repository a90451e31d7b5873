#!/usr/bin/env python3
"""Incremental cost of LM passes in the pair-mode axle kernel: kernel time vs max_iter for several batch sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem

def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for k in (2, 32, 90):
    program, targets = axle_grid_problem(k, k)
    n = targets.shape[0]
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device="cuda:0")
    info = torch.empty((n, 40), dtype=torch.uint8, device="cuda:0")
    row = [timed(dp.plan(t, out=out, info_out=info, chain_len=1, max_iter=m)) for m in (1, 2, 3, 4, 5, 100)]
    print(f"B={n:6d}  us for max_iter=1,2,3,4,5,100: " + " ".join(f"{v:8.2f}" for v in row))
