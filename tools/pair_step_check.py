#!/usr/bin/env python3
"""Pair-mode kernels, first LM step from the design state (max_iter = 1, own first pass) against numpy on the oracle's
Jacobian: cost at the design state (all rows incl. the joining ones) and the damped, coupled step.
   python3 tools/pair_step_check.py <fixture> ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
from oracle.oracle import Oracle
np.set_printoptions(linewidth=220, precision=4, suppress=True)
for name in sys.argv[1:]:
    arrays, program = load_golden(name)
    program = program.with_line_mode("pinned")
    dp = DeviceProgram(program, "cuda:0")
    n = program.n_vars
    targets = arrays["targets_abs"].reshape(-1, program.n_targets)[:8]
    x0 = program.design_pos[program.free_point].reshape(-1)
    xs = np.repeat(x0[None], len(targets), 0)
    r_o, jac_o = Oracle(program).eval(xs, targets)
    ata = np.einsum("bij,bik->bjk", jac_o, jac_o)
    atr = np.einsum("bij,bi->bj", jac_o, r_o)
    lam0 = dp.default_opts().lambda0
    res = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain_len=1, max_iter=1, shared_first_step=False, predictor=False)
    torch.cuda.synchronize()
    info = res.info()
    free_out = [list(program.out_point).index(p) for p in program.free_point]
    got = res.positions.cpu().numpy()[:, free_out].reshape(-1, n) - xs
    print(name, "kernel", dp.kernel, "n", n)
    fp = [str(program.point_keys[p]) for p in program.free_point]
    for b in range(len(targets)):
        lam = lam0 * np.max(np.diag(ata[b]))
        want = -np.linalg.solve(ata[b] + lam * np.eye(n), atr[b])
        err = np.abs(got[b] - want)
        worst = np.argsort(-err)[:4]
        print(f"  problem {b}: |want| {np.abs(want).max():.3e}  max err {err.max():.3e} at vars {worst} ({[fp[w // 3] for w in worst]}); "
              f"nfev {info['nfev'][b]} iters {info['iterations'][b]} flags {info['flags'][b]}")
