#!/usr/bin/env python3
"""LM trace of one problem of a pair-mode fixture (cold, own first pass) beside the oracle's cost at the design state."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
from oracle.oracle import Oracle
np.set_printoptions(linewidth=200, precision=4)
name, prob = sys.argv[1], int(sys.argv[2])
arrays, program = load_golden(name)
program = program.with_line_mode("pinned")
dp = DeviceProgram(program, "cuda:0")
targets = arrays["targets_abs"].reshape(-1, program.n_targets)
x0 = program.design_pos[program.free_point].reshape(-1)
r_o, jac_o = Oracle(program).eval(x0[None], targets[prob:prob + 1])
print("oracle cost at design", 0.5 * float(r_o[0] @ r_o[0]), "max|r|", np.abs(r_o).max(), "row", int(np.abs(r_o[0]).argmax()), "of", r_o.shape[1])
t = torch.as_tensor(targets, device="cuda:0")
tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), prob)
res = dp.solve(t, chain_len=1, predictor=False, shared_first_step=False, max_iter=6)
torch.cuda.synchronize()
a = tr.cpu().numpy()
print("pass: mode Ft Fc lambda step rho accept done")
for k in range(0, 12):
    if np.any(a[k] != 0): print("  ", k, a[k])
