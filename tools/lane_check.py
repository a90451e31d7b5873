#!/usr/bin/env python3
"""Lane kernel (one lane per problem) against the quad kernel: the pass quantities (r, J^T J, J^T r, damped step), the
solved positions, the evaluation counts and the time per launch on the grid / ensemble shapes it is meant for.
   python3 tools/lane_check.py [c4|c5|c2] [grid edge / geometries]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd import _lib
_lib.LIB_PATH = os.environ.get("LANE_TL_LIB", _lib.LIB_PATH)   # (A/B against another build of the library)
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram


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
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda", 0)
    kw = {}
    if which == "c5":
        program, table, rel = W.ensemble_problem(size or 4096, 256)
        dp = DeviceProgram(program, dev)
        gpos, gparam = dp.rebind(torch.as_tensor(table, device=dev))
        targets = dp.ensemble_targets(gpos, rel)
        kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=rel.shape[0])
    elif which == "c4":
        program, t = W.macpherson_grid_problem(size or 512, size or 512)
        dp = DeviceProgram(program, dev)
        targets = torch.as_tensor(t, device=dev)
    else:
        program, t = W.bump_sweep_problem(size or 16384)
        dp = DeviceProgram(program, dev)
        targets = torch.as_tensor(t, device=dev)
    print(f"{which}: {targets.shape[0]} problems, kernel={dp.kernel!r}, lane threshold {dp.lane_threshold}, lane note {dp.lane_note!r}")
    if dp.lane_threshold < 0:
        return 1
    # the pass: r, J^T J, J^T r, dx of both generated kernels at perturbed free vectors (own geometry)
    rng = np.random.default_rng(0)
    x0 = program.design_pos[program.free_point].reshape(-1)
    x = torch.as_tensor(x0[None] + rng.normal(0.0, 3.0, (256, program.n_vars)), device=dev)
    tt = torch.as_tensor(np.asarray(program_targets(program, x0))[None].repeat(256, 0) + rng.normal(0.0, 2.0, (256, program.n_targets)), device=dev)
    for lam in (0.0, 1e-3):
        q = dp.quad_eval(x, tt, lam)
        l = dp.quad_eval(x, tt, lam, lane=True)
        names = ("r", "ata", "atr", "dx")
        print(f"  lambda={lam}: " + ", ".join(f"max|{n}_lane - {n}_quad| = {float((a - b).abs().max()):.2e} (scale {float(b.abs().max()):.1e})"
                                         for n, a, b in zip(names, l, q)))
    for mode, cl in (("cold", 1), ("chained", -1)):
        rq = dp.solve(targets, chain_len=cl, kernel="quad", predictor=False, **kw)
        rl = dp.solve(targets, chain_len=cl, kernel="lane", predictor=False, **kw)
        torch.cuda.synchronize()
        iq, il = rq.info(), rl.info()
        d = float((rq.positions - rl.positions).abs().max())
        print(f"  {mode}: max|lane - quad| = {d:.2e} mm; converged lane {int(((il['flags'] & 7) == 1).sum())} quad {int(((iq['flags'] & 7) == 1).sum())} of {len(il)}; "
              f"nfev lane {il['nfev'].mean():.3f} quad {iq['nfev'].mean():.3f}; max residual lane {il['max_residual'].max():.2e}")
        n = targets.shape[0]
        out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
        info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
        for kern in ("quad", "lane"):
            launch = dp.plan(targets, out=out, info_out=info, chain_len=cl, predictor=False, kernel=kern, **kw)
            ms = timed(launch)
            print(f"    {kern:5s} {mode:8s}: {ms:.4f} ms  {n / ms / 1e3:.1f} M solves/s")
    return 0


def program_targets(program, x0):
    """The design state's own target values (target rows evaluated at the design positions)."""
    vals = []
    for t in range(program.n_targets):
        p = program.tgt_point[t]
        vals.append(float(np.dot(program.design_pos[p], program.tgt_dir[t])))
    return vals


if __name__ == "__main__":
    sys.exit(main())
