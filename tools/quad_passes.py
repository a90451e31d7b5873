#!/usr/bin/env python3
"""Incremental cost of LM passes in the quad kernel: kernel time vs max_iter for several batch sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for n in (16, 1024, 16384, 65536):
    program, targets = bump_sweep_problem(n)
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device="cuda:0")
    info = torch.empty((n, 40), dtype=torch.uint8, device="cuda:0")
    row = []
    for k in (1, 2, 3, 4, 5, 100):
        row.append(timed(dp.plan(t, out=out, info_out=info, chain_len=1, max_iter=k)))
    print(f"B={n:6d}  us for max_iter=1,2,3,4,5,100: " + " ".join(f"{v:7.2f}" for v in row))
