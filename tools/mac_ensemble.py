#!/usr/bin/env python3
"""A MacPherson sensitivity ensemble (perturbed hardpoints x bump sweep, per-geometry tables): lane kernel time per launch.
   python3 tools/mac_ensemble.py [geometries] [steps]      (LANE_TL_LIB=<other libokx.so> for an A/B pair)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd import _lib
_lib.LIB_PATH = os.environ.get("LANE_TL_LIB", _lib.LIB_PATH)
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import macpherson_grid_problem

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
program, _ = macpherson_grid_problem(2, 2)
dp = DeviceProgram(program, dev)
rng = np.random.default_rng(0)
table = np.repeat(program.design_pos[None], G, axis=0)
table[1:] += rng.normal(0.0, 0.5, size=(G - 1,) + program.design_pos.shape)
gpos, gparam = dp.rebind(torch.as_tensor(table, device=dev))
rel = np.stack([np.zeros(S), np.linspace(-50.0, 60.0, S)], axis=1)
t = dp.ensemble_targets(gpos, rel)
kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=S, chain_len=1, predictor=False)
launch = dp.plan(t, **kw)
for _ in range(5):
    res = launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
inf = res.info()
print(f"{G} geometries x {S} steps = {G * S} problems: {ms * 1e3:.1f} us per launch = {G * S / ms * 1e3:.3e} solves/s, nfev {inf['nfev'].mean():.3f}, "
      f"converged {float(np.mean((inf['flags'] & 7) == 1)):.4f}, kernel {dp.kernel}, lane from {dp.lane_threshold}")
