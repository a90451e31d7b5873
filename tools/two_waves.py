#!/usr/bin/env python3
"""One against two wavefronts per SIMD for the quad kernels (developer switch quad_two_waves: __launch_bounds__(64, 2), i.e.
at most 256 registers per lane) on launches of one and of several rounds, independent cold starts, kernel = quad:
   python3 tools/two_waves.py            (profiles/r05/EXPERIMENTS.md section 4)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem, macpherson_grid_problem

dev = torch.device("cuda:0")
cases = [("dw 16384 (one round)", lambda: bump_sweep_problem(16384)), ("dw 65536", lambda: bump_sweep_problem(65536)),
         ("dw 262144", lambda: bump_sweep_problem(262144)), ("macpherson 512 x 512", lambda: macpherson_grid_problem(512, 512))]
rows = []
for name, make in cases:
    program, t = make()
    targets = torch.as_tensor(t, device=dev)
    n = targets.shape[0]
    row = {"case": name, "problems": n}
    ref = None
    for tag, switch in (("one_wave", ""), ("two_waves", "quad_two_waves")):
        os.environ["OKX_DEV"] = switch
        dp = DeviceProgram(program, dev)
        out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
        info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
        launch = dp.plan(targets, out=out, info_out=info, chain_len=1, predictor=False, kernel="quad")
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        _, ms = bench.time_launches(launch, 50, 5, dev)
        nfev, ok = bench.info_summary(info)
        row[tag] = {"kernel_ms": ms, "solves_per_s": n / ms * 1e3, "lm_evaluations_mean": nfev, "all_converged": ok}
        if ref is None:
            ref = out.clone()
        else:
            row["max_abs_difference_mm"] = float((out - ref).abs().max())
        dp.close()
    row["two_over_one"] = row["one_wave"]["kernel_ms"] / row["two_waves"]["kernel_ms"]
    rows.append(row)
    print(f"{name:24s} one wave/SIMD {row['one_wave']['kernel_ms']:.4f} ms  two {row['two_waves']['kernel_ms']:.4f} ms  speed-up {row['two_over_one']:.3f}  "
          f"max |d| {row['max_abs_difference_mm']:.1e}", file=sys.stderr)
os.environ["OKX_DEV"] = ""
print(json.dumps({"rows": rows}))
