#!/usr/bin/env python3
"""Edge-of-reach exploration: walk one target outward from the design state until the device flags the step, then
compare device and oracle (MINPACK on J, QR) at fractions of that reach.  tools/reach.py [dw|mac|axle]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
from oracle.oracle import Oracle

which = sys.argv[1] if len(sys.argv) > 1 else "dw"
program, targets = {"dw": lambda: W.bump_sweep_problem(4), "mac": lambda: W.macpherson_grid_problem(2, 2),
                    "axle": lambda: W.axle_grid_problem(2, 2)}[which]()
base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
dp = DeviceProgram(program, "cuda:0")
orc = Oracle(program)
print(which, "n", program.n_vars, "targets", program.target_desc)

def walk(direction, far, steps=2048):
    s = np.linspace(0.0, far, steps)
    t = base[None] + s[:, None] * direction[None]
    res = dp.solve(torch.as_tensor(t, device="cuda:0"), chain=True)
    info = res.info(); ok = res.accepted(info) & (info["max_residual"] <= 1e-5)   # true reach: the target rows are still met
    bad = np.nonzero(~ok)[0]
    return s, (s[bad[0] - 1] if bad.size else None), info, res

for name, direction in (("t0+", [1, 0, 0]), ("t0-", [-1, 0, 0]), ("t1+", [0, 1, 0]), ("t1-", [0, -1, 0])):
    direction = np.array(direction[: program.n_targets] + [0] * max(0, program.n_targets - 3), dtype=float)[: program.n_targets]
    if which == "axle" and name.startswith("t1"):
        direction = np.array([1.0, -1.0, 0.0]) * (1 if name.endswith("+") else -1)   # roll
    if which == "axle" and name.startswith("t0"):
        direction = np.array([1.0, 1.0, 0.0]) * (1 if name.endswith("+") else -1)    # heave
    s, reach, info, res = walk(direction, 600.0)
    if reach is None:
        print(name, "no lock-out within 600 mm"); continue
    # refine
    s2, reach2, _, _ = walk(direction, reach + 600.0 / 2047 * 1.5, 4096)
    reach = reach2 if reach2 is not None else reach
    print(f"{name}: reach {reach:.3f} mm")
    for frac in (0.99, 0.999, 0.9999, 1.0, 1.0001):
        path = np.linspace(0.0, frac * reach, 257)
        t = base[None] + path[:, None] * direction[None]
        res = dp.solve(torch.as_tensor(t, device="cuda:0"), chain=True); inf = res.info()
        cold = dp.solve(torch.as_tensor(t[-1:], device="cuda:0")); cinf = cold.info()
        o = orc.sweep(t, 1e-15, 1e-15, 1e-15, warm_start=True)
        pos = res.positions.cpu().numpy()
        err = np.abs(pos[-1] - o.positions[-1]).max()
        cerr = np.abs(cold.positions.cpu().numpy()[0] - o.positions[-1]).max()
        r, j = orc.eval(o.x[-1], t[-1]); sv = np.linalg.svd(j[0], compute_uv=False)
        def newton_gap(xfree):
            rr, jj = orc.eval(xfree, t[-1]); return float(np.abs(np.linalg.lstsq(jj[0], -rr[0], rcond=None)[0]).max())
        fo = [int(np.nonzero(program.out_point == p)[0][0]) for p in program.free_point]
        gap_o = newton_gap(o.x[-1]); gap_d = newton_gap(pos[-1][fo].reshape(-1))
        # polished truth: Gauss-Newton (SVD lstsq) from the oracle's point
        xt = o.x[-1].copy()
        for _ in range(8):
            rr, jj = orc.eval(xt, t[-1]); xt = xt + np.linalg.lstsq(jj[0], -rr[0], rcond=None)[0]
        err_true = np.abs(pos[-1][fo].reshape(-1) - xt).max(); err_o_true = np.abs(o.x[-1] - xt).max()
        print(f"   {frac:7.4f}: chained err {err:.2e} flags {inf['flags'][-1]} nfev {inf['nfev'][-1]} mres {inf['max_residual'][-1]:.1e} | cold err {cerr:.2e} flags {cinf['flags'][0]} nfev {cinf['nfev'][0]}"
              f" | gap dev {gap_d:.1e} orc {gap_o:.1e} | vs polished: dev {err_true:.1e} orc {err_o_true:.1e}" f" | oracle ok {o.first_failed_step} minpack {o.info['minpack_info'][-1]} nfev {o.info['nfev'][-1]} mres {o.info['max_residual'][-1]:.1e} | cond(J) {sv[0]/sv[-1]:.2e}")
