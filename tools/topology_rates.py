#!/usr/bin/env python3
"""What every reference geometry fixture gets: kernel family and solves/s for a 16384-problem sweep, cold and chained.
The sweep is the fixture's own golden sweep (tests/golden/t_*.npz and the BASELINE ones) stretched to 16384 steps by linear
interpolation between its first and last target rows (a line through the fixture's reachable range).
   python3 tools/topology_rates.py > profiles/r03/topology_rates.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram

NAMES = ["c1_dw_corner", "c4_macpherson_grid", "t_corner_strut", "t_corner_rocker", "t_corner_strut_rocker", "t_axle_macpherson",
         "t_axle_dw", "c3_axle_grid", "t_axle_t_bar_roll", "t_axle_heave_link", "t_axle_t_bar_heave"]
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows = []
for name in NAMES:
    arrays, program = load_golden(name)
    program = program.with_line_mode("pinned")
    t = arrays["targets_abs"].reshape(-1, program.n_targets)
    lo, hi = t.min(axis=0), t.max(axis=0)
    # a line from the all-low corner to the all-high corner of the fixture's target box, shrunk by 10 % at both ends
    s = np.linspace(0.1, 0.9, n)[:, None]
    targets = torch.as_tensor(lo + s * (hi - lo), device=dev)
    dp = DeviceProgram(program, dev)
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
    info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
    row = {"fixture": name, "n_vars": program.n_vars, "rows": program.n_residuals, "kernel": dp.kernel,
           "why_not_generated": dp.kernel_note or None}
    for tag, cl in (("cold", 1), ("chained", -1)):
        launch = dp.plan(targets, out=out, info_out=info, chain_len=cl, predictor=False)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        reps = 50 if dp.kernel == "quad" else 5
        wall, ms = bench.time_launches(launch, reps, 2, dev)
        nfev, ok = bench.info_summary(info)
        row[tag] = {"solves_per_s": n / (ms * 1e-3), "kernel_ms": ms, "lm_evaluations_mean": nfev, "all_converged": ok}
    rows.append(row)
    print(f"{name:24s} n={program.n_vars:3d} {dp.kernel:5s} cold {row['cold']['solves_per_s'] / 1e6:9.2f} M/s ({row['cold']['lm_evaluations_mean']:.2f} ev)  "
          f"chained {row['chained']['solves_per_s'] / 1e6:9.2f} M/s  ok={row['cold']['all_converged'] and row['chained']['all_converged']}", file=sys.stderr)
    dp.close()
print(json.dumps({"problems": n, "rows": rows}, indent=1))
