import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
np.set_printoptions(linewidth=200, precision=4)
for prob in (0, 100, 4000, 8192, 8200, 12000, 16383):
    tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda:0")
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), prob)
    res = dp.solve(t, chain_len=1, kernel="quad", predictor=False)
    torch.cuda.synchronize()
    a = tr.cpu().numpy()
    print("problem", prob, "target", targets[prob], ": pass mode Ft Fc lambda step rho accept done")
    for k in range(0, 8):
        if a[k].any(): print("  ", k, a[k])
