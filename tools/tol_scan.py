import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem
dev = "cuda:0"
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for name, make in (("C2", lambda: bump_sweep_problem(16384)), ("C3", lambda: axle_grid_problem(256, 256)), ("C4", lambda: macpherson_grid_problem(512, 512))):
    p, t = make()
    dp = DeviceProgram(p, dev)
    tt = torch.as_tensor(t, device=dev)
    tight = dp.solve(tt, chain_len=1, predictor=False, step_tol=1e-13, confirm_full_pass=True, max_iter=200).positions.clone()
    for tol in (1e-11, 3e-11, 1e-10, 3e-10, 1e-9):
        launch = dp.plan(tt, chain_len=1, predictor=False, step_tol=tol)
        ms = timed(launch)
        res = launch(); torch.cuda.synchronize()
        inf = res.info()
        print(f"{name} step_tol {tol:7.0e}: {ms*1e3:8.2f} us, nfev {inf['nfev'].mean():.3f}, max |x - tight| {float((res.positions - tight).abs().max()):.2e}, converged {bool(np.all((inf['flags'] & 7) == 1))}", flush=True)
    dp.close()
