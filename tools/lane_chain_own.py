#!/usr/bin/env python3
"""Own-geometry double-wishbone sweeps on the lane kernel: independent solves against chains (auto length) for sweeps of
262144 ... 1048576 steps, records and no position stores."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram
dev = torch.device("cuda", 0)
for n in (65536, 262144, 1048576):
    program, t = W.bump_sweep_problem(n)
    dp = DeviceProgram(program, dev)
    targets = torch.as_tensor(t, device=dev)
    info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
    for cl in (1, -1):
        row = []
        for mode in ("records", "none"):
            out = None if mode == "none" else torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
            launch = dp.plan(targets, out=out, info_out=info, chain_len=cl, output=mode, predictor=False, kernel="lane")
            for _ in range(5): launch()
            wall, ms = bench.time_launches(launch, 20, 3, dev)
            row.append(f"{mode} {ms:.4f} ms")
        print(f"n={n:8d} chain_len={cl:2d}: " + "  ".join(row) + f"  (evaluations {bench.info_summary(info)[0]:.3f})")
    dp.close()
