#!/usr/bin/env python3
"""LM trajectory (cost after k iterations) of the quad kernel vs the generic kernel."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

program, targets = bump_sweep_problem(16)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
for k in range(1, 9):
    for kern in ("single", "quad"):
        res = dp.solve(t, chain_len=1, kernel=kern, max_iter=k)
        i = res.info()
        print(k, kern, "cost", i["cost"][[0, 7, 15]], "nfev", i["nfev"][[0, 7, 15]], "it", i["iterations"][[0, 7, 15]],
              "step", i["last_step"][[0, 7, 15]], "flags", i["flags"][[0, 7, 15]])
