#!/usr/bin/env python3
"""Where one round of the lane kernel (65536 independent double-wishbone solves, one wavefront per SIMD) spends its time:
the launch cut short on the shared first step, max_iter 1 / 2 / 3, with records / without position stores."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "dw"
n = 65536
program, t = W.bump_sweep_problem(n) if which == "dw" else W.macpherson_grid_problem(256, 256)
dp = DeviceProgram(program, dev)
tg = torch.as_tensor(t, device=dev)
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
for label, kw in (("full solve", {}), ("ends on the shared first step", {"step_tol": 1e9}), ("max_iter 1", {"max_iter": 1}),
                  ("max_iter 2", {"max_iter": 2}), ("max_iter 3", {"max_iter": 3}), ("own first pass (no table)", {"shared_first_step": False})):
    row = []
    for mode in ("records", "none"):
        launch = dp.plan(tg, out=out if mode == "records" else None, info_out=info, chain_len=1, predictor=False, kernel="lane", output=mode, **kw)
        for _ in range(50): launch()
        wall, ms = bench.time_launches(launch, 300, 10, dev)
        row.append(f"{mode} {ms * 1e3:6.2f} us")
    print(f"{which} {label:32s}: " + "  ".join(row) + f"  evaluations {bench.info_summary(info)[0]:.2f}")
