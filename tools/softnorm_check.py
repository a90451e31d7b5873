#!/usr/bin/env python3
"""Every axle fixture with the reference's literal point-on-line rows (line_mode = softnorm) on its generated kernel:
converged, and where against the reference's tight run."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
for name in sys.argv[1:] or ["c3_axle_grid", "t_axle_dw", "t_axle_macpherson", "t_axle_t_bar_roll", "t_axle_t_bar_bump", "t_axle_heave_link", "t_axle_t_bar_heave", "t_corner_strut_rocker"]:
    arrays, program = load_golden(name)  # softnorm rows, as flattened from the reference
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(arrays["targets_abs"].reshape(-1, program.n_targets), device="cuda:0")
    res = dp.solve(t, chain_len=1, step_tol=1e-8, max_iter=200)
    i = res.info()
    d = np.abs(res.positions.cpu().numpy() - arrays["ref_tight_pos"].reshape(len(t), -1, 3)).max()
    print(f"{name:24s} {dp.kernel:5s} line_mode={program.line_mode}: accepted {int(res.accepted(i).sum())}/{len(t)}, nfev mean {i['nfev'].mean():.1f} max {i['nfev'].max()}, "
          f"max |device - reference tight| = {d:.2e} mm, max residual {i['max_residual'].max():.2e}")
