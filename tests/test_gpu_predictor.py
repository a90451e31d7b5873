"""GPU: chain-head predictor (okx_program_fit_predictor): same solutions, fewer LM evaluations."""

import numpy as np
import pytest
import torch

from conftest import gpu_available

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _solve_both(program, targets, **kw):
    from open_kinematics_amd.batch import DeviceProgram

    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    cold = dp.solve(t, predictor=False, **kw)
    pred = dp.solve(t, predictor=True, **kw)
    torch.cuda.synchronize()
    return dp, t, cold, pred


@pytest.mark.parametrize("workload", ["dw", "mac"])
def test_predictor_gives_the_same_states_in_fewer_evaluations(workload):
    from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem

    program, targets = {"dw": lambda: bump_sweep_problem(4096), "mac": lambda: macpherson_grid_problem(48, 48),
                        "axle": lambda: axle_grid_problem(24, 24)}[workload]()
    dp, t, cold, pred = _solve_both(program, targets, chain_len=1)
    assert dp.kernel == "quad" and dp.predictor_box is not None
    ci, pi = cold.info(), pred.info()
    assert cold.accepted(ci).all() and pred.accepted(pi).all()
    assert float((cold.positions - pred.positions).abs().max()) <= 1e-9  # north-star tolerance
    # (a cold start with the second-order shared first step is itself down to ~3 evaluations: the model's margin is one)
    assert pi["nfev"].mean() <= ci["nfev"].mean() - 0.5
    assert pi["nfev"].max() <= ci["nfev"].max()
    # a chained launch: heads (and their successors) start from the model, later steps from the secant
    chained = dp.solve(t, chain_len=8, predictor=True)
    assert float((cold.positions - chained.positions).abs().max()) <= 1e-9
    every = dp.solve(t, chain_len=8, predictor="all")
    assert float((cold.positions - every.positions).abs().max()) <= 1e-9


def test_pair_mode_programs_go_without():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import axle_grid_problem

    program, targets = axle_grid_problem(8, 8)
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    assert not dp.fit_predictor(t) and "pair-mode" in dp._predictor_note
    assert dp.solve(t).accepted(dp.solve(t).info()).all()  # auto: plain starts


def test_targets_outside_the_fitted_box_are_clamped_not_extrapolated():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(512)  # -60 .. +80 mm
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    inner = t[(t[:, 1] > t[:, 1].median() - 10.0) & (t[:, 1] < t[:, 1].median() + 10.0)]
    assert dp.fit_predictor(inner, required=True)
    lo, hi = dp.predictor_box
    assert hi[1] - lo[1] < 25.0 and lo[0] == hi[0]  # the rack target is held: not a fitted dimension
    cold = dp.solve(t, chain_len=1, predictor=False)
    pred = dp.solve(t, chain_len=1, predictor=True)  # most of the sweep lies outside the box
    assert pred.accepted(pred.info()).all()
    assert float((cold.positions - pred.positions).abs().max()) <= 1e-9
    assert dp.fit_predictor(t, degree=9, required=True)  # refit over the whole sweep
    again = dp.solve(t, chain_len=1, predictor=True)
    assert float((cold.positions - again.positions).abs().max()) <= 1e-9
    assert again.info()["nfev"].mean() < pred.info()["nfev"].mean()


def test_predictor_is_skipped_where_it_does_not_apply(golden):
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(64)
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    with pytest.raises(RuntimeError, match="not fitted"):
        dp.fit_predictor(required=True)
    # generic interpreter kernel: no predictor, plain solve
    a = dp.solve(t, kernel="single")
    assert dp._predictor is None
    # a plain solve never fits a model behind the caller's back (the fit is a synchronous step), whatever the batch size ...
    small = dp.solve(t)
    assert dp._predictor is None and float((small.positions - a.positions).abs().max()) <= 1e-9
    big_program, big_targets = bump_sweep_problem(4096)
    big = DeviceProgram(big_program, "cuda:0")
    plain = big.solve(torch.as_tensor(big_targets, device="cuda:0"))
    assert big._predictor is None and big.predictor_box is None
    # ... but uses one that was fitted explicitly
    assert big.fit_predictor(torch.as_tensor(big_targets, device="cuda:0"))
    modelled = big.solve(torch.as_tensor(big_targets, device="cuda:0"))
    assert big._predictor is True and modelled.info()["nfev"].mean() < plain.info()["nfev"].mean()
    assert float((modelled.positions - plain.positions).abs().max()) <= 1e-9
    # per-geometry launches never use it
    gpos, gparam = dp.rebind(torch.as_tensor(program.design_pos[None]))
    dp.fit_predictor(t, required=True)
    b = dp.solve(t, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=64, predictor=True)
    assert float((a.positions - b.positions).abs().max()) <= 1e-9
    # an unsolvable node (far outside the mechanism's reach): no model, launches fall back to cold starts
    far = t.clone()
    far[0, 1] += 5000.0
    assert not dp.fit_predictor(far)
    assert "did not converge" in dp._predictor_note
    c = dp.solve(t)
    assert float((a.positions - c.positions).abs().max()) <= 1e-9
    with pytest.raises(RuntimeError, match="no chain-head predictor"):
        dp.solve(t, predictor=True)
    assert not dp.fit_predictor(t, degree=13) and "degree" in dp._predictor_note


def test_two_varying_targets_on_the_double_wishbone_against_the_oracle():
    """Rack x bump grid on the DW corner: the 2-D model (36 terms), chains with row wraps and cold starts agree
    with each other and with the oracle's MINPACK on a sample."""
    from oracle.oracle import Oracle
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, base = bump_sweep_problem(2)
    rack = np.linspace(-25.0, 25.0, 96)
    bump = np.linspace(-70.0, 90.0, 128)
    t0 = base[0] - np.array([0.0, -60.0])  # design targets (bump_sweep_problem starts at -60 mm)
    grid = np.stack(np.meshgrid(rack, bump, indexing="ij"), -1).reshape(-1, 2) + t0
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(grid, device="cuda:0")
    cold = dp.solve(t, chain_len=1, predictor=False)
    assert cold.accepted(cold.info()).all()
    assert dp.fit_predictor(t, required=True)
    lo, hi = dp.predictor_box
    assert hi[0] - lo[0] == pytest.approx(50.0) and hi[1] - lo[1] == pytest.approx(160.0)
    for kw in (dict(chain_len=1, predictor=True), dict(chain_len=-1, predictor=True), dict(chain_len=128, predictor=False),
               dict(chain_len=1000, predictor="all")):
        res = dp.solve(t, **kw)
        assert res.accepted(res.info()).all(), kw
        assert float((res.positions - cold.positions).abs().max()) <= 1e-9, kw
    model = dp.solve(t, chain_len=1, predictor=True).info()
    assert model["nfev"].mean() <= cold.info()["nfev"].mean() - 0.9
    pick = np.linspace(0, grid.shape[0] - 1, 40).astype(int)
    orc = Oracle(program).sweep(grid[pick], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert orc.first_failed_step == -1
    assert np.max(np.abs(cold.positions.cpu().numpy()[pick] - orc.positions)) <= 1e-9
