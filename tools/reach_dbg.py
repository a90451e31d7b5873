import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_gpu_reach as T
from open_kinematics_amd.batch import DeviceProgram
from oracle.oracle import Oracle
program, directions = T._problem("axle")
base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
dp = DeviceProgram(program, "cuda:0"); orc = Oracle(program)
for name in ("heave-", "heave+", "roll+"):
    direction = np.asarray(directions[name], dtype=np.float64)
    reach, met = T._reach(dp, base, direction)
    print(name, "reach", reach, "met", met)
    for frac in (0.9999, 1.0, 1.0001, 1.001):
        path = np.linspace(0.0, frac * reach, 257)
        targets = base[None] + path[:, None] * direction[None]
        oracle = orc.sweep(targets, 1e-15, 1e-15, 1e-15, warm_start=True)
        r, j = orc.eval(oracle.x[-1], targets[-1]); sv = np.linalg.svd(j[0], compute_uv=False)
        x_true, gap = T._polish(orc, oracle.x[-1], targets[-1])
        ch = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain=True).info()
        co = dp.solve(torch.as_tensor(targets[-1:], device="cuda:0")).info()
        print(f"  {frac}: cond(J) {sv[0]/sv[-1]:.2e} smin {sv[-1]:.2e} gap {gap:.1e} |r|max {np.abs(r).max():.1e} chained flags {ch['flags'][-1]} mres {ch['max_residual'][-1]:.1e} nfev {ch['nfev'][-1]} cold flags {co['flags'][0]} mres {co['max_residual'][0]:.1e}")
print("--- tangent pivots at the solved states (undamped J^T J, pair mode: regularised halves)")
for name in ("heave-", "heave+"):
    direction = np.asarray(directions[name], dtype=np.float64)
    reach, met = T._reach(dp, base, direction)
    for frac in (1.0, 1.0001):
        path = np.linspace(0.0, frac * reach, 257)
        targets = base[None] + path[:, None] * direction[None]
        res = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain=True)
        tan, tinfo = dp.tangents(res.positions[-1:].contiguous())
        ti = dp.tangent_info(tinfo)
        inf = res.info()
        print(name, frac, "piv_lo(last_step)", inf["last_step"][-1], "piv_hi(cost)", inf["cost"][-1], "tangent min/max pivot", ti["min_pivot"], ti["max_pivot"], "flags", ti["flags"], "| solve flags", inf["flags"][-1], "iters", inf["iterations"][-1], "last_step", inf["last_step"][-1], "cost", inf["cost"][-1])
