#!/usr/bin/env python3
"""Throughput of every BASELINE configuration at full size on one GPU (not the headline bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W

def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out

KERNELS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto"]

def report(name, dp, targets, **kw):
    # start of a chain head: "design" = cold start from the design state, "model" = fitted Chebyshev predictor
    for kern in KERNELS:
        for cl in (1, -1):
            for start in ("design", "model"):
                if start == "model" and ("geom_pos" in kw or kern not in ("auto", "quad") or not dp.fit_predictor(targets)):
                    continue  # per-geometry launches, interpreter kernels and pair-mode programs have no predictor
                b = targets.shape[0]
                out = torch.empty((b, dp.program.n_out, 3), dtype=torch.float64, device="cuda")
                info_out = torch.empty((b, 40), dtype=torch.uint8, device="cuda")
                dt, res = timed(dp.plan(targets, chain_len=cl, kernel=kern, out=out, info_out=info_out,
                                        predictor=start == "model", **kw))
                info = res.info()
                ok = bool(np.all((info["flags"] & 7) == 1))
                print(f"{name:34s} {kern:6s} chain_len={cl:2d} start={start:6s}  B={targets.shape[0]:8d}  {targets.shape[0]/dt/1e6:8.3f} M solves/s  "
                      f"{dt*1e3:8.3f} ms  evals {info['nfev'].mean():.2f}  max_res {info['max_residual'].max():.2e}  converged={ok}")

p, t = W.bump_sweep_problem(16384)
report("C2 DW corner 16384-step bump", DeviceProgram(p), torch.as_tensor(t, device="cuda"))
for steps in (65536, 262144, 1048576):
    p, t = W.bump_sweep_problem(steps)
    report(f"   DW corner {steps}-step bump", DeviceProgram(p), torch.as_tensor(t, device="cuda"))
p, t = W.macpherson_grid_problem(512, 512)
report("C4 MacPherson 512x512 bump x rack", DeviceProgram(p), torch.as_tensor(t, device="cuda"))
p, t = W.axle_grid_problem(256, 256)
report("C3 rocker axle 256x256 heave x roll", DeviceProgram(p), torch.as_tensor(t, device="cuda"))
p, table, rel = W.ensemble_problem(4096, 256)
dp = DeviceProgram(p)
t0 = time.perf_counter()
gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda")); torch.cuda.synchronize()
print(f"C5 rebind of 4096 geometries: {(time.perf_counter()-t0)*1e3:.2f} ms")
base = torch.stack([gpos[:, p.tgt_point[k]] @ torch.as_tensor(p.tgt_dir[k], device="cuda") for k in range(p.n_targets)], 1)  # [G,T]
targets = (base[:, None, :] + torch.as_tensor(rel, device="cuda")[None]).reshape(-1, p.n_targets).contiguous()
report("C5 4096 geometries x 256 steps", dp, targets, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=256)
