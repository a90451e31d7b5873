#!/usr/bin/env python3
"""Where the C2 cold sweep's time goes: the same launch with the solve cut short.
step_tol = 1e9 ends every problem on the shared first step (prologue + record stores + launch gap only);
max_iter = 1, 2, 3 allow that many factorisations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
dev = torch.device("cuda:0")
p, t = W.bump_sweep_problem(16384)
dp = DeviceProgram(p, dev)
tg = torch.as_tensor(t, device=dev)
out = torch.empty((tg.shape[0], p.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((tg.shape[0], 40), dtype=torch.uint8, device=dev)
for label, kw in (("full solve", {}), ("ends on the shared first step", {"step_tol": 1e9}),
                  ("max_iter 1", {"max_iter": 1}), ("max_iter 2", {"max_iter": 2}), ("max_iter 3", {"max_iter": 3}),
                  ("own first pass, ends on it", {"step_tol": 1e9, "shared_first_step": False}),
                  ("256 problems only", {"n": 256}), ("256 problems, first step only", {"n": 256, "step_tol": 1e9})):
    n = kw.pop("n", tg.shape[0])
    launch = dp.plan(tg[:n], out=out[:n], info_out=info[:n], chain_len=1, predictor=False, **kw)
    wall, ms = bench.time_launches(launch, 200, 10, dev)
    nfev, ok = bench.info_summary(info[:n])
    print(f"{label:34s}: {1e3 * ms:7.2f} us per launch, evaluations {nfev:.2f}")
