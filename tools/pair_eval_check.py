#!/usr/bin/env python3
"""Pair-mode kernels: what the generated code computes at seeded points (r, J^T r, damped step) against the oracle / numpy.
   python3 tools/pair_eval_check.py <fixture> ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
from oracle.oracle import Oracle
np.set_printoptions(linewidth=220, precision=3)
for name in sys.argv[1:]:
    arrays, program = load_golden(name)
    program = program.with_line_mode("pinned")
    dp = DeviceProgram(program, "cuda:0")
    x, t = arrays["eval_x"], arrays["eval_targets"]
    r_o, jac_o = Oracle(program).eval(x, t)
    ata_o = np.einsum("bij,bik->bjk", jac_o, jac_o)
    atr_o = np.einsum("bij,bi->bj", jac_o, r_o)
    lam = 1e-6 * float(np.max(np.diagonal(ata_o, axis1=1, axis2=2)))
    r, ata, atr, dx = [v.cpu().numpy() for v in dp.quad_eval(x, t, lam)]
    n = program.n_vars
    dx_o = np.stack([-np.linalg.solve(ata_o[k] + lam * np.eye(n), atr_o[k]) for k in range(len(x))])
    print(name, "kernel", dp.kernel, "n", n, "m", program.n_residuals)
    print("  max|r - r_o|", np.abs(r - r_o).max(), "rows worst", np.argsort(-np.abs(r - r_o).max(0))[:6], "of", r.shape[1])
    print("  max|atr - atr_o|", np.abs(atr - atr_o).max(), "scale", np.abs(atr_o).max(), "vars worst", np.argsort(-np.abs(atr - atr_o).max(0))[:6])
    print("  max|dx - dx_o|", np.abs(dx - dx_o).max(), "scale", np.abs(dx_o).max(), "vars worst", np.argsort(-np.abs(dx - dx_o).max(0))[:8])
    fp = [program.point_keys[p] for p in program.free_point]
    print("  free points", [str(k) for k in fp])
