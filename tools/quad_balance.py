#!/usr/bin/env python3
"""How evenly a full launch's wave units spread over its wavefronts: C3 (256 x 256 axle grid, pair-mode cold body: 8192 units on
1024 wavefronts) from a build generated with OKX_DEV=quad_timeline - every wavefront stamps the device's real-time counter at
entry and after each of its units (the last one stays).  Prints when the wavefronts end relative to the launch: if the last one
ends long after the median, the launch waits for a few unlucky wavefronts and a work queue would pay.
   OKX_DEV=quad_timeline,no_lane OKX_KERNEL_CACHE=build/kc_tl python3 tools/quad_balance.py [c3 | c2x8]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem

what = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
if what == "c3":
    program, targets = axle_grid_problem(256, 256)
    per_wave = 8
else:  # eight rounds of the single-mode cold body
    program, targets = bump_sweep_problem(8 * 16384)
    per_wave = 16
n = targets.shape[0]
dp = DeviceProgram(program, dev)
t = torch.as_tensor(targets, device=dev)
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
units = (n + per_wave - 1) // per_wave
waves = min(units, 1024)
tr = torch.zeros((2 * waves, 16), dtype=torch.float64, device=dev)
dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), -1)
launch = dp.plan(t, out=out, info_out=info, chain_len=1, predictor=False, kernel="quad")
for _ in range(20):
    launch()
torch.cuda.synchronize()
tr.zero_()
launch()
torch.cuda.synchronize()
both = tr.cpu().numpy()
dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
sec = both[waves:]
rt0, rt1 = sec[:, 0], sec[:, 1]
assert (rt0 > 0).all() and (rt1 > 0).all(), "no real-time stamps: is this a quad_timeline build, and did the cold body run?"
base = rt0.min()
end = (rt1 - base) / 100.0  # us
busy = (rt1 - rt0) / 100.0
inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=[("max_residual", "f8"), ("cost", "f8"), ("last_step", "f8"), ("iterations", "i4"),
                                                          ("nfev", "i4"), ("flags", "i4"), ("reserved", "i4")])
print(f"{what}: {n} problems, {units} wave units on {waves} wavefronts ({units / waves:.1f} each); nfev mean {inf['nfev'].mean():.3f}")
print(f"wavefronts end (us after the first one starts): min {end.min():.1f}, 5 % {np.percentile(end, 5):.1f}, median {np.median(end):.1f}, "
      f"95 % {np.percentile(end, 95):.1f}, last {end.max():.1f}")
print(f"busy time per wavefront: mean {busy.mean():.1f} us, max {busy.max():.1f}; a perfectly even launch would end at ~{busy.mean() + (rt0 - base).mean() / 100.0:.1f} us: "
      f"the launch is {end.max() / (busy.mean() + (rt0 - base).mean() / 100.0):.3f} x that")
