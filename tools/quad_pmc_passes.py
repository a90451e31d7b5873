"""Dispatches for rocprofv3 --pmc: the C2 quad solve with max_iter = 1 .. 5 (3 launches each, in that order)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
for k in (1, 2, 3, 4, 5):
    for _ in range(3):
        dp.solve(t, chain_len=1, max_iter=k)
torch.cuda.synchronize()
