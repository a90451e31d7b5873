#!/usr/bin/env python3
"""Kernel time of the pair-mode fixtures at 16384 problems (cold / chained), for A/B of generator switches:
   OKX_DEV=1 OKX_KERNEL_CACHE=... python3 tools/pair_rates.py [fixture ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
dev = torch.device("cuda", 0)
n = 16384
for name in sys.argv[1:] or ["c3_axle_grid", "t_axle_t_bar_roll"]:
    arrays, program = load_golden(name)
    program = program.with_line_mode("pinned")
    t = arrays["targets_abs"].reshape(-1, program.n_targets)
    lo, hi = t.min(axis=0), t.max(axis=0)
    targets = torch.as_tensor(lo + np.linspace(0.1, 0.9, n)[:, None] * (hi - lo), device=dev)
    dp = DeviceProgram(program, dev)
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
    info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
    row = []
    for cl in (1, -1):
        launch = dp.plan(targets, out=out, info_out=info, chain_len=cl, predictor=False)
        for _ in range(20): launch()
        wall, ms = bench.time_launches(launch, 100, 5, dev)
        row.append(f"{ms * 1e3:8.1f} us ({bench.info_summary(info)[0]:.2f} ev)")
    print(f"{name:22s} cold {row[0]}  chained {row[1]}")
