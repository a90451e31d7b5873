#!/usr/bin/env python3
"""Shared first step on / off at the full BASELINE sizes: same answers, evaluation counts (a run, not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
def report(name, dp, targets, **kw):
    for cl in (1, -1):
        a = dp.solve(targets, chain_len=cl, predictor=False, shared_first_step=False, **kw)
        b = dp.solve(targets, chain_len=cl, predictor=False, shared_first_step=True, **kw)
        ia, ib = a.info(), b.info()
        print(f"{name} chain_len={cl}: max |own - shared| = {float((a.positions - b.positions).abs().max()):.2e} mm, accepted {a.accepted(ia).all()} / {b.accepted(ib).all()}, "
              f"evals {ia['nfev'].mean():.3f} -> {ib['nfev'].mean():.3f}, ill-conditioned flags {int((ib['flags'] & 8 != 0).sum())}")
p, t = W.bump_sweep_problem(16384); report("C2", DeviceProgram(p, "cuda:0"), torch.as_tensor(t, device="cuda:0"))
p, t = W.macpherson_grid_problem(512, 512); report("C4", DeviceProgram(p, "cuda:0"), torch.as_tensor(t, device="cuda:0"))
p, table, rel = W.ensemble_problem(4096, 256, sigma=2.0, seed=9)
dp = DeviceProgram(p, "cuda:0")
gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda:0"))
report("C5 (sigma 2 mm)", dp, dp.ensemble_targets(gpos, rel), geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=256)
