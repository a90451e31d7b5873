#!/usr/bin/env python3
"""C3 (rocker + U-bar axle grid, pair mode): kernel time and evaluation counts, cold and chained, with and without the
shared first step (OKX_PAIR_NO_HEAD=1 generates the pair-mode kernel without the table).
   python3 tools/c3_ab.py [grid edge]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
program, targets = axle_grid_problem(k, k)
dp = DeviceProgram(program, "cuda:0")
n = targets.shape[0]
t = torch.as_tensor(targets, device="cuda:0")
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device="cuda:0")
info = torch.empty((n, 40), dtype=torch.uint8, device="cuda:0")
print(f"kernel {dp.kernel!r}; OKX_PAIR_NO_HEAD={os.environ.get('OKX_PAIR_NO_HEAD')}; shares first step: {dp.shares_first_step}")
ref = dp.solve(t, chain_len=1, shared_first_step=False).positions.clone()
for cl in (1, -1):
    for shared in (False, True):
        res = dp.solve(t, chain_len=cl, shared_first_step=shared)
        i = res.info()
        d = float((res.positions - ref).abs().max())
        ms = timed(dp.plan(t, out=out, info_out=info, chain_len=cl, shared_first_step=shared))
        print(f"  chain_len={cl:2d} shared_first_step={shared!s:5}: {ms:.4f} ms  {n / ms / 1e3:7.1f} M/s  nfev {i['nfev'].mean():.3f} "
              f"converged {int(((i['flags'] & 7) == 1).sum())}/{n}  max|d|={d:.1e}")
