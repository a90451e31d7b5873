#!/usr/bin/env python3
"""Cold solves near the edge of reach (tests/test_gpu_reach.py's hardest case): MacPherson, rack+ direction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_reach as T
from open_kinematics_amd.batch import DeviceProgram
which = sys.argv[1] if len(sys.argv) > 1 else "mac"
program, directions = T._problem(which)
base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
dp = DeviceProgram(program, "cuda:0")
for name, direction in directions.items():
    direction = np.asarray(direction, dtype=np.float64)
    reach, met = T._reach(dp, base, direction)
    fr = np.array([0.5, 0.8, 0.9, 0.95, 0.99, 0.999, 0.9999])
    t = base[None] + (fr * reach)[:, None] * direction[None]
    chained = dp.solve(torch.as_tensor(base[None] + np.linspace(0, reach, 513)[:, None] * direction[None], device="cuda:0"), chain=True)
    for sfs in (True, False):
        cold = dp.solve(torch.as_tensor(t, device="cuda:0"), chain_len=1, shared_first_step=sfs)
        i = cold.info()
        print(f"{name} reach {reach:.2f} shared_first_step={sfs}: flags {i['flags'].tolist()} nfev {i['nfev'].tolist()} iters {i['iterations'].tolist()}")
