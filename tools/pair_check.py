#!/usr/bin/env python3
"""Pair-mode quad kernel (one quad per axle half) vs the generic wavefront kernel on the rocker axle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
program, targets = axle_grid_problem(k, k)
t0 = time.time()
dp = DeviceProgram(program, "cuda:0")
print(f"create {time.time()-t0:.1f}s kernel={dp.kernel!r} note={dp.kernel_note!r}")
t = torch.as_tensor(targets, device="cuda:0")
ref = dp.solve(t, chain_len=1, kernel="single")
iref = ref.info()
print("wave: converged", int(((iref["flags"] & 7) == 1).sum()), "/", len(iref), "nfev", iref["nfev"].mean())
if dp.kernel == "quad":
    for cl in (1, -1):
        res = dp.solve(t, chain_len=cl, kernel="quad")
        iq = res.info()
        d = (res.positions - ref.positions).abs().max().item()
        print(f"quad chain_len={cl}: converged {int(((iq['flags'] & 7) == 1).sum())}/{len(iq)} nfev {iq['nfev'].mean():.3f} max {iq['nfev'].max()} "
              f"max|quad-wave|={d:.3e} maxres {iq['max_residual'].max():.2e} vs wave maxres {iref['max_residual'].max():.2e}")
        bad = np.nonzero((iq["flags"] & 7) != 1)[0]
        if len(bad): print("  first bad", bad[:4], iq[bad[:4]])
    for cl in (1, -1):
        for kern in ("single", "quad"):
            ms = timed(lambda: dp.solve(t, chain_len=cl, kernel=kern))
            print(f"  {kern:6s} chain_len={cl:2d}: {ms:.3f} ms  {len(targets)/ms/1e3:.2f} M solves/s")
