#!/usr/bin/env python3
"""Initial damping (lambda0, relative to max diag J^T J) against evaluations / time / answers, cold and chained."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
dev = torch.device("cuda:0")
for name, make in (("C3", lambda: W.axle_grid_problem(256, 256)), ("C2", lambda: W.bump_sweep_problem(16384)), ("C4", lambda: W.macpherson_grid_problem(512, 512))):
    p, t = make()
    dp = DeviceProgram(p, dev)
    tg = torch.as_tensor(t, device=dev)
    out = torch.empty((tg.shape[0], p.n_out, 3), dtype=torch.float64, device=dev)
    info = torch.empty((tg.shape[0], 40), dtype=torch.uint8, device=dev)
    ref = None
    for cl in (1, -1):
        for lam in (1e-6, 1e-7, 1e-8, 1e-10, 0.0):
            launch = dp.plan(tg, out=out, info_out=info, chain_len=cl, predictor=False, lambda0=lam)
            wall, ms = bench.time_launches(launch, 10, 2, dev)
            nfev, ok = bench.info_summary(info)
            if ref is None: ref = out.clone()
            print(f"{name} chain_len={cl:2d} lambda0={lam:7.0e}: kernel {ms:.4f} ms evals {nfev:.3f} ok {ok} vs first {float((out - ref).abs().max()):.1e}")
