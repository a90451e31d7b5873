#!/usr/bin/env python3
"""C5 on the lane kernel: independent solves against chains, with full records / free coordinates / no position stores
(what the per-lane record stores of a chain step cost)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram
dev = torch.device("cuda", 0)
program, table, rel = W.ensemble_problem(4096, 256)
dp = DeviceProgram(program, dev)
gpos, gparam = dp.rebind(torch.as_tensor(table, device=dev))
targets = dp.ensemble_targets(gpos, rel)
kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=rel.shape[0], predictor=False, kernel="lane")
n = targets.shape[0]
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
for cl in (1, -1, 2, 8, 16):
    row = []
    for mode in ("records", "free", "none"):
        out = None if mode == "none" else torch.empty((n, program.n_out if mode == "records" else program.n_free, 3), dtype=torch.float64, device=dev)
        launch = dp.plan(targets, out=out, info_out=info, chain_len=cl, output=mode, **kw)
        for _ in range(5): launch()
        wall, ms = bench.time_launches(launch, 20, 3, dev)
        row.append(f"{mode} {ms:.4f} ms")
    print(f"chain_len={cl:3d}: " + "  ".join(row) + f"  (evaluations {bench.info_summary(info)[0]:.3f})")
