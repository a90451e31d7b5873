import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(101, line_mode="softnorm")
dp = DeviceProgram(program, "cuda:0")
print(dp.kernel, dp.kernel_note)
t = torch.as_tensor(targets, device="cuda:0")
for kern in ("single", "quad"):
    res = dp.solve(t, chain_len=1, kernel=kern, step_tol=1e-8, max_iter=200)
    i = res.info()
    bad = np.nonzero((i["flags"] & 1) == 0)[0]
    print(kern, "nfev mean", i["nfev"].mean(), "max", i["nfev"].max(), "bad", bad, i[bad])
    print("  iterations", i["iterations"][40:50], "nfev", i["nfev"][40:50])
