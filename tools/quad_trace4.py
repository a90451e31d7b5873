import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(101, line_mode="softnorm")
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
sel = [75]
for k in list(range(12, 60, 1)):
    for kern in ("quad",):
        res = dp.solve(t, chain_len=1, kernel=kern, max_iter=k, step_tol=1e-8)
        i = res.info()
        print(k, kern, "cost", i["cost"][sel], "nfev", i["nfev"][sel], "it", i["iterations"][sel], "step", i["last_step"][sel], "flags", i["flags"][sel], "mres", i["max_residual"][sel])
