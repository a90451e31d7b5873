"""
Edge of reach (promoted from tools/stress.py / tools/reach.py): one target is walked outward from the design state
until the device flags the step (the reference's "did not reach an acceptable residual", ``core/solver.py:726-747``:
kinematic lock-out), then the device is compared with the oracle's MINPACK — QR on J, no normal equations — at 90,
95, 99, 99.9, 99.99 and 100 % of the TRUE reach (the last step whose target rows are still met: max residual at the
softnorm floor), warm-started along the path like the reference AND as one cold start from the design state, and
just beyond it (100.01 .. 101 %), where the reference still accepts compromise points (max residual under its 1e-3
tolerance) although the minimiser is singular there (cond(J) ~ 1e10).  The device solves the normal equations (LDL^T of J^T J), whose condition number is cond(J)^2, so this
is where it would lose digits first.  The yardstick is MINPACK's answer POLISHED by Gauss-Newton steps with an SVD
least-squares solve (the correction then sits at its rounding floor, <= 1e-10 mm): within a few 1e-4 mm of a singular configuration
MINPACK's own xtol stop leaves it up to 1e-8 mm short (measured with tools/reach.py), the polish does not.
Contract: every problem is within 1e-9 mm of that point or carries a flag (not converged / residual exceeded /
OKX_INFO_ILL_CONDITIONED); where the polish itself cannot converge (singular J) the record must show it: a flag, or a
worst residual above the floor of a state that meets its targets.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOL = 1e-9  # mm, north_star tolerance
FRACTIONS = (0.9, 0.95, 0.99, 0.999, 0.9999, 1.0)
BEYOND = (1.0001, 1.001, 1.01)
TARGETS_MET = 2.0  # x the design state's own max residual (the softnorm floor, ~1e-6 mm): the target rows are still met


def _problem(which):
    from open_kinematics_amd import workloads as W

    if which == "dw":
        return W.bump_sweep_problem(4)[0], {"rack+": [1, 0], "rack-": [-1, 0], "bump+": [0, 1], "bump-": [0, -1]}
    if which == "mac":
        return W.macpherson_grid_problem(2, 2)[0], {"rack+": [1, 0], "rack-": [-1, 0], "bump+": [0, 1], "bump-": [0, -1]}
    return W.axle_grid_problem(2, 2)[0], {"heave+": [1, 1, 0], "heave-": [-1, -1, 0], "roll+": [1, -1, 0], "roll-": [-1, 1, 0],
                                          "rack+": [0, 0, 1]}


def _polish(orc, x, targets):
    """Gauss-Newton with an SVD least-squares solve from MINPACK's point; (x, last correction)."""
    x = np.array(x, dtype=np.float64)
    step = np.inf
    for _ in range(12):
        r, jac = orc.eval(x, targets)
        dx = np.linalg.lstsq(jac[0], -r[0], rcond=None)[0]
        step = float(np.abs(dx).max())
        if not np.isfinite(step) or step > 1.0:
            return x, np.inf
        x = x + dx
        if step <= 1e-13:  # (the rounding floor of the correction itself is ~|J^+| * 1e-13: 1e-12 .. 1e-11 near the edge)
            break
    return x, step


def _reach(dp, base, direction, far=700.0):
    """Largest displacement along `direction` at which the chained device sweep still meets its target rows (two passes);
    also the residual level that means "met"."""
    lo, hi = 0.0, far
    floor = None
    for steps in (1024, 1024):
        s = np.linspace(lo, hi, steps)
        if lo > 0.0:  # the fine pass still walks out from the design state, densely near the edge
            s = np.concatenate([np.linspace(0.0, lo, 256, endpoint=False), s])
        res = dp.solve(torch.as_tensor(base[None] + s[:, None] * direction[None], device=dp.device), chain=True)
        info = res.info()
        if floor is None:
            floor = TARGETS_MET * float(info["max_residual"][0])  # s[0] = 0: the design state
        ok = res.accepted(info) & (info["max_residual"] <= floor)
        bad = np.nonzero(~ok)[0]
        if bad.size == 0:
            return None, floor
        lo, hi = s[bad[0] - 1], s[bad[0]]
    return lo, floor


@pytest.mark.parametrize("which", ["dw", "mac", "axle"])
def test_every_problem_near_lock_out_is_accurate_or_flagged(which):
    from open_kinematics_amd._abi import INFO_CONVERGED, INFO_FAILED, INFO_ILL_CONDITIONED, INFO_RESIDUAL_EXCEEDED
    from open_kinematics_amd.batch import DeviceProgram
    from oracle.oracle import Oracle

    program, directions = _problem(which)
    base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
    dp = DeviceProgram(program, "cuda:0")
    orc = Oracle(program)
    checked = flagged = 0
    for name, direction in directions.items():
        direction = np.asarray(direction, dtype=np.float64)
        reach, met = _reach(dp, base, direction)
        assert reach is not None and reach > 20.0, f"{which} {name}: no lock-out found inside 700 mm"
        for frac in FRACTIONS + BEYOND:
            path = np.linspace(0.0, frac * reach, 257)
            targets = base[None] + path[:, None] * direction[None]
            oracle = orc.sweep(targets, 1e-15, 1e-15, 1e-15, warm_start=True)
            assert frac > 1.0 or oracle.first_failed_step == -1, f"{which} {name} {frac}: the oracle itself rejects the path"
            x_true, gap = _polish(orc, oracle.x[-1], targets[-1])
            defined = gap <= 1e-8   # beyond that J is singular to working precision: no yardstick, a flag is due
            tol = max(TOL, 10.0 * gap) if defined else TOL  # the yardstick's own rounding floor, reached only at the very edge
            assert (defined and tol == TOL) or frac >= 1.0, f"{which} {name} {frac}: no well-defined minimiser inside the reach"
            want = orc.positions(x_true)[program.out_point] if defined else oracle.positions[-1]
            chained = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain=True)
            cold = dp.solve(torch.as_tensor(targets[-1:], device="cuda:0"))
            for label, res, row in (("chained", chained, -1), ("cold", cold, 0)):
                flags = int(res.info()["flags"][row])
                err = float(np.abs(res.positions[row].cpu().numpy() - want).max())
                is_flagged = (flags & INFO_CONVERGED) == 0 or (flags & (INFO_RESIDUAL_EXCEEDED | INFO_FAILED | INFO_ILL_CONDITIONED)) != 0
                checked += 1
                flagged += is_flagged
                if defined:
                    assert err <= tol or is_flagged, f"{which} {name} {frac} {label}: {err:.2e} mm from the polished oracle, flags {flags}"
                else:
                    # singular compromise point beyond lock-out: the record must carry a flag - in pair mode too, where the
                    # conditioning test sees the halves' pivots AND the stiffness of the mode the joining row ties together
                    worst = float(res.info()["max_residual"][row])
                    assert is_flagged, \
                        f"{which} {name} {frac} {label}: singular configuration (correction {gap:.1e}) not flagged: flags {flags}, max residual {worst:.1e}"
                if frac <= 0.999:  # inside the reach nothing may hide behind a flag
                    assert err <= TOL and not is_flagged, f"{which} {name} {frac} {label}: {err:.2e} mm, flags {flags}"
    assert checked == 2 * len(FRACTIONS + BEYOND) * len(directions)
    assert flagged <= 2 * (len(BEYOND) + 1) * len(directions)  # only at the edge and beyond
    dp.close()
