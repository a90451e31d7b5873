import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=5, suppress=True)
a = dp.solve(t, chain_len=1, kernel="single", max_iter=1).positions.cpu().numpy()
b = dp.solve(t, chain_len=1, kernel="quad", max_iter=1).positions.cpu().numpy()
print("design"); print(program.design_pos)
print("single - design, problem 0"); print(a[0] - program.design_pos)
print("quad - design, problem 0"); print(b[0] - program.design_pos)
