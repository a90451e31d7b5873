"""
The shared first step (okx_solve_opts.shared_first_step): the Levenberg-Marquardt pass AT the design
state is evaluated once per geometry and every chain head takes its first step from that table.  Same iteration: the
results must equal the ones of problems that run their own first pass, to the solver's tolerance, with one evaluation
less per cold start — for the program's own geometry, for per-geometry tables, in every chain mode, for both line
modes, and against the oracle.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _both(dp, targets, **kw):
    own = dp.solve(targets, shared_first_step=False, **kw)
    shared = dp.solve(targets, shared_first_step=True, **kw)
    torch.cuda.synchronize()
    return own, shared


@pytest.mark.parametrize("workload", ["dw", "mac", "axle"])
@pytest.mark.parametrize("line_mode", ["pinned", "softnorm"])
def test_cold_starts_take_the_same_path_with_one_evaluation_less(workload, line_mode, monkeypatch, tmp_path):
    from open_kinematics_amd.batch import DeviceProgram

    from open_kinematics_amd import workloads as W
    from oracle.oracle import Oracle

    program, targets = {"dw": lambda: W.bump_sweep_problem(1024, line_mode), "mac": lambda: W.macpherson_grid_problem(32, 32, line_mode),
                        "axle": lambda: W.axle_grid_problem(24, 24, line_mode)}[workload]()  # axle: pair-mode kernel, 3 program targets
    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad"
    t = torch.as_tensor(targets, device="cuda:0")
    # the reference's zero-gradient line row converges linearly and is only defined to ~3e-8 mm:
    # looser step tolerance and more iterations, as in test_softnorm_rows_on_device_reach_the_same_point
    extra = {} if line_mode == "pinned" else dict(step_tol=1e-8, max_iter=200)
    own, shared = _both(dp, t, chain_len=1, predictor=False, **extra)
    io, ish = own.info(), shared.info()
    assert own.accepted(io).all() and shared.accepted(ish).all()
    tol = 1e-9 if line_mode == "pinned" else 6e-8
    assert float((own.positions - shared.positions).abs().max()) <= tol
    # the own path minus its first evaluation - and, since the table carries the second-order terms of the step
    # (round 3), usually one more: the trial point is second-order accurate in the target displacement
    saved = io["nfev"].astype(int) - ish["nfev"].astype(int)
    if line_mode == "pinned" and workload != "axle":
        assert np.median(saved) >= 1 and saved.min() >= 0 and np.mean(saved >= 1) >= 0.95
        assert np.mean(ish["iterations"] <= io["iterations"]) >= 0.95
    elif line_mode == "pinned":  # the coupled halves take marginal accept / reject decisions differently now and then
        assert float(np.mean(saved)) >= 0.5 and saved.min() >= -2
    else:  # dozens of linearly converging passes along the valley: rounding decides the exact count
        assert abs(float(np.mean(saved)) - 1.0) <= 1.0
    if line_mode == "pinned":
        pick = np.linspace(0, len(targets) - 1, 24).astype(int)
        orc = Oracle(program).sweep(targets[pick], 1e-15, 1e-15, 1e-15, warm_start=False)
        assert float(np.abs(shared.positions.cpu().numpy()[pick] - orc.positions).max()) <= 1e-9


def test_design_targets_need_no_iteration_but_still_land_on_the_minimiser():
    """A problem whose targets ARE the design values: the table's constraint-gradient column carries the softnorm offset
    of the distance rows (-1e-6 each), so the answer is the true minimiser (1e-6 mm off the design state), not the
    design state itself."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, _ = bump_sweep_problem(4)
    base = np.array([float(program.design_pos[p] @ d) for p, d in zip(program.tgt_point, program.tgt_dir)])
    t = torch.as_tensor(np.tile(base, (16, 1)), device="cuda:0")
    dp = DeviceProgram(program, "cuda:0")
    own, shared = _both(dp, t, chain_len=1, predictor=False)
    assert shared.accepted(shared.info()).all()
    assert float((own.positions - shared.positions).abs().max()) <= 1e-10


def test_chains_and_geometry_tables():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import ensemble_problem

    program, table, rel = ensemble_problem(48, 64, sigma=1.0, seed=3)
    dp = DeviceProgram(program, "cuda:0")
    gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda:0"))
    targets = dp.ensemble_targets(gpos, rel)
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=64, predictor=False)
    ref = None
    for chain_len in (1, 8, -1, 0):
        own, shared = _both(dp, targets, chain_len=chain_len, chain=chain_len == 0, **kw)
        assert own.accepted(own.info()).all() and shared.accepted(shared.info()).all()
        assert float((own.positions - shared.positions).abs().max()) <= 1e-9
        heads = 48 * 64 if chain_len == 1 else 48 * (64 // 8 if chain_len == 8 else 1)
        saved = int(own.info()["nfev"].sum()) - int(shared.info()["nfev"].sum())
        assert 0.9 * heads <= saved <= 2.2 * heads or chain_len == -1  # one or two evaluations per chain head
        ref = shared.positions if ref is None else ref
        assert float((ref - shared.positions).abs().max()) <= 1e-9
    # a second, larger ensemble through the same program: the scratch table grows, the results stay right
    program2, table2, rel2 = ensemble_problem(160, 16, sigma=1.0, seed=4)
    gpos2, gparam2 = dp.rebind(torch.as_tensor(table2, device="cuda:0"))
    t2 = dp.ensemble_targets(gpos2, rel2)
    a, b = _both(dp, t2, chain_len=1, geom_pos=gpos2, geom_row_param=gparam2, steps_per_geometry=16, predictor=False)
    assert float((a.positions - b.positions).abs().max()) <= 1e-9 and b.accepted(b.info()).all()


def test_lambda0_changes_refresh_the_cached_table():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(256)
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    for lam in (1e-6, 1e-3, 1e-6):
        own, shared = _both(dp, t, chain_len=1, predictor=False, lambda0=lam)
        assert shared.accepted(shared.info()).all()
        assert float((own.positions - shared.positions).abs().max()) <= 1e-9
