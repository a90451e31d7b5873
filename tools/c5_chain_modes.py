#!/usr/bin/env python3
"""BASELINE config 5 (4096 geometries x 256 steps) on the lane kernel: independent cold starts against warm-started chains
of a few lengths (the flat chain body), time per launch, evaluations, and the largest difference from the cold answers.
   python3 tools/c5_chain_modes.py [chain lengths ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import ensemble_problem

dev = torch.device("cuda:0")
lengths = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]
program, table, rel = ensemble_problem()
dp = DeviceProgram(program, dev)
gpos, grow = dp.rebind(torch.as_tensor(table, device=dev))
targets = dp.ensemble_targets(gpos, rel)
n = targets.shape[0]
kw = dict(geom_pos=gpos, geom_row_param=grow, steps_per_geometry=rel.shape[0], predictor=False)
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
rows = []
ref = None
for tag, extra in [("cold", dict(chain_len=1))] + [(f"chain {k}", dict(chain_len=k, kernel="lane")) for k in lengths] + [("auto", dict(chain_len=-1))]:
    launch = dp.plan(targets, out=out, info_out=info, **extra, **kw)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    _, ms = bench.time_launches(launch, 10, 2, dev)
    nfev, ok = bench.info_summary(info)
    if ref is None:
        ref = out.clone()
    diff = float((out - ref).abs().max())
    rows.append({"mode": tag, "kernel_ms": ms, "solves_per_s": n / ms * 1e3, "lm_evaluations_mean": nfev, "all_converged": ok, "max_abs_difference_from_cold_mm": diff})
    print(f"{tag:10s} {ms:.4f} ms  {n / ms * 1e3:.3g} solves/s  {nfev:.2f} evaluations  ok={ok}  max |d| vs cold {diff:.1e}", file=sys.stderr)
print(json.dumps({"rows": rows}))
