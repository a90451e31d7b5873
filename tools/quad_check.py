#!/usr/bin/env python3
"""Quad (runtime-specialised) kernel vs the generic wavefront kernel and the CPU oracle: agreement and timing."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem, macpherson_grid_problem


def timed(fn, reps=10):
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    which = sys.argv[2] if len(sys.argv) > 2 else "dw"
    if which == "dw":
        program, targets = bump_sweep_problem(n)
    else:
        k = int(round(n ** 0.5))
        program, targets = macpherson_grid_problem(k, k)
    t0 = time.time()
    dp = DeviceProgram(program, "cuda:0")
    print(f"program create {time.time() - t0:.2f} s; kernel={dp.kernel!r} note={dp.kernel_note!r}")
    t = torch.as_tensor(targets, device="cuda:0")
    ref = dp.solve(t, chain_len=1, kernel="single")
    torch.cuda.synchronize()
    iref = ref.info()
    print("wave kernel: converged", int(((iref["flags"] & 7) == 1).sum()), "/", len(iref), "nfev mean", iref["nfev"].mean())
    if dp.kernel != "quad":
        return
    for cl in (1, -1):
        res = dp.solve(t, chain_len=cl, kernel="quad")
        torch.cuda.synchronize()
        iq = res.info()
        d = (res.positions - ref.positions).abs().max().item()
        print(f"quad chain_len={cl}: converged {int(((iq['flags'] & 7) == 1).sum())}/{len(iq)} nfev mean {iq['nfev'].mean():.3f} "
              f"max {iq['nfev'].max()} max|quad-wave|={d:.3e} maxres {iq['max_residual'].max():.2e}")
        bad = np.nonzero((iq["flags"] & 7) != 1)[0]
        if len(bad):
            print("  first bad", bad[:5], iq[bad[:5]])
    for cl in (1, -1):
        for kern in ("single", "quad"):
            ms = timed(lambda: dp.solve(t, chain_len=cl, kernel=kern))
            print(f"  {kern:6s} chain_len={cl:2d}: {ms:.4f} ms  {len(targets) / ms / 1e3:.1f} M solves/s")
    if len(targets) <= 2048:
        from oracle.oracle import Oracle
        orc = Oracle(program).sweep(targets, 1e-15, 1e-15, 1e-15, warm_start=False)
        res = dp.solve(t, chain_len=1, kernel="quad")
        print("max|quad - oracle| =", float(np.max(np.abs(res.positions.cpu().numpy() - orc.positions))))


if __name__ == "__main__":
    main()
