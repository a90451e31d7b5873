"""Where the device and the oracle's MINPACK differ most on the C4 grid (diagnostic for tests/test_gpu_fullsize.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import macpherson_grid_problem
from oracle.oracle import Oracle
from test_gpu_fullsize import _stratified

program, targets = macpherson_grid_problem(512, 512)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(targets, device="cuda:0")
pick = _stratified(targets.shape[0], 4, 1024, seed=4)
res = dp.solve(t[pick].contiguous(), chain_len=1, predictor=False)
pos = res.positions.cpu().numpy()
tan, tinfo = dp.tangents(res.positions)
ti = dp.tangent_info(tinfo)
orc = Oracle(program)
ref = orc.sweep(targets[pick], 1e-15, 1e-15, 1e-15, warm_start=False)
err = np.abs(pos - ref.positions).reshape(len(pick), -1).max(axis=1)
order = np.argsort(-err)[:12]
base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
free_out = [list(program.out_point).index(int(p)) for p in program.free_point]
for k in order:
    x_dev = pos[k][free_out].reshape(1, -1)
    r_dev, _ = orc.eval(x_dev, targets[pick[k]][None], jac=False)
    r_ref, _ = orc.eval(ref.x[k][None], targets[pick[k]][None], jac=False)
    print(f"idx {pick[k]:7d} rel targets {targets[pick[k]] - base} err {err[k]:.3e} cond(JtJ) {ti['max_pivot'][k] / ti['min_pivot'][k]:.3e} "
          f"|r|dev {np.abs(r_dev).max():.2e} |r|ref {np.abs(r_ref).max():.2e} nfev_ref {ref.info['nfev'][k]} dev last_step {res.info()['last_step'][k]:.1e}")
print("quantiles of err", np.quantile(err, [0.5, 0.9, 0.99, 0.999, 1.0]))
c = ti['max_pivot'] / ti['min_pivot']
print("corr log err vs log cond", np.corrcoef(np.log10(err + 1e-16), np.log10(c))[0, 1])

# who is off: polish both answers with Gauss-Newton steps on the oracle's own r / J (float64) and see which one moves
for k in order[:6]:
    tt = targets[pick[k]][None]
    for tag, x0 in (("dev", pos[k][free_out].reshape(-1)), ("ref", ref.x[k].copy())):
        x = x0.copy()
        moves = []
        for it in range(4):
            r, J = orc.eval(x[None], tt, jac=True)
            r, J = r[0], J[0]
            dx = -np.linalg.lstsq(J, r, rcond=None)[0]
            x = x + dx
            moves.append(float(np.abs(dx).max()))
        print(f"idx {pick[k]:7d} {tag}: GN moves {['%.1e' % m for m in moves]} total |x - x0| {np.abs(x - x0).max():.2e}")
