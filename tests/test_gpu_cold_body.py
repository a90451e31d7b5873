"""
The quad kernel's cold body (okx_quad_cold_u: tables staged through LDS, the all-accepted passes in a fast loop that hands
over in place to the general loop, staged stores) against its general body (okx_quad_solve_u) on independent solves.  The two
run the same evaluation, factorisation and update formulas; the cold body only organises them differently - so the answers,
the flags and the evaluation counts must be the SAME, bit for bit, also where the fast loop has to hand over (rejected steps
beyond the reach) and for ragged batch sizes, compact outputs and pair mode.  Parity of either body with the oracle and the
reference is what tests/test_gpu_parity.py, test_gpu_quad.py and test_gpu_fullsize.py establish.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

INFO = [("max_residual", "f8"), ("cost", "f8"), ("last_step", "f8"), ("iterations", "i4"), ("nfev", "i4"), ("flags", "i4"), ("reserved", "i4")]


def _both_bodies(program, targets, monkeypatch, output="records"):
    from open_kinematics_amd.batch import DeviceProgram

    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad" and dp.has_cold_body, dp.kernel_note
    t = torch.as_tensor(targets, device="cuda:0")
    results = []
    for switch in (None, "no_cold"):
        if switch:
            monkeypatch.setenv("OKX_DEV", switch)
        else:
            monkeypatch.delenv("OKX_DEV", raising=False)
        res = dp.solve(t, chain_len=1, predictor=False, kernel="quad", output=output)
        torch.cuda.synchronize()
        pos = (res.positions if output == "records" else res.free).cpu().numpy()
        results.append((pos, np.frombuffer(res.info_raw.cpu().numpy().tobytes(), dtype=INFO).copy()))
    monkeypatch.delenv("OKX_DEV", raising=False)
    dp.close()
    return results


def _assert_same(cold, general):
    (pc, ic), (pg, ig) = cold, general
    assert np.array_equal(ic["flags"], ig["flags"]) and np.array_equal(ic["nfev"], ig["nfev"])
    assert np.array_equal(ic["iterations"], ig["iterations"])
    ok = (ig["flags"] & 7) == 1
    assert np.array_equal(pc[ok], pg[ok])                       # bit for bit where the solve converged
    assert np.array_equal(ic["max_residual"][ok], ig["max_residual"][ok]) and np.array_equal(ic["cost"][ok], ig["cost"][ok])
    return ok


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 16384])
def test_double_wishbone_sweeps_of_any_length(monkeypatch, n):
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(n)
    ok = _assert_same(*_both_bodies(program, targets, monkeypatch))
    assert ok.all()


@pytest.mark.parametrize("output", ["free", "none"])
def test_compact_outputs(monkeypatch, output):
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(1000)
    if output == "none":
        dp = DeviceProgram(program, "cuda:0")
        t = torch.as_tensor(targets, device="cuda:0")
        a = dp.solve(t, chain_len=1, predictor=False, output="none").info_raw.cpu().numpy()
        monkeypatch.setenv("OKX_DEV", "no_cold")
        b = dp.solve(t, chain_len=1, predictor=False, output="none").info_raw.cpu().numpy()
        assert np.array_equal(a, b)
        dp.close()
        return
    cold, general = _both_bodies(program, targets, monkeypatch, output="free")
    assert _assert_same(cold, general).all()


def test_grids_beyond_the_reach_hand_over(monkeypatch):
    """Targets out to and beyond the edge of the reach: rejected steps, stops on the cost test, failures - the fast loop hands
    its wavefronts to the general loop, which must go on exactly as the general body does.  A double-wishbone bump x rack
    grid (single mode) and an axle heave x roll grid (pair mode)."""
    from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem

    program, base = bump_sweep_problem(2)
    bump, rack = np.meshgrid(np.linspace(-260.0, 260.0, 64), np.linspace(-120.0, 120.0, 32), indexing="ij")
    t = np.stack([base[0, 0] + rack.ravel(), 0.5 * (base[0, 1] + base[1, 1]) + bump.ravel()], axis=1)
    cold, general = _both_bodies(program, t, monkeypatch)
    ok = _assert_same(cold, general)
    assert 0.1 < ok.mean() < 0.95                               # the grid really reaches past the mechanism's limits
    assert (general[1]["nfev"] > 6).any()                        # ... and some solves really needed rejected steps
    program, base = axle_grid_problem(2, 2)
    centre = 0.5 * (base[0] + base[-1])
    heave, roll = np.meshgrid(np.linspace(-75.0, 75.0, 32), np.linspace(-45.0, 45.0, 32), indexing="ij")
    t = np.stack([centre[0] + (heave + roll).ravel(), centre[1] + (heave - roll).ravel(), np.full(heave.size, centre[2])], axis=1)
    cold, general = _both_bodies(program, t, monkeypatch)
    ok = _assert_same(cold, general)
    assert 0.3 < ok.mean() < 1.0


def test_pair_mode_axle_grid(monkeypatch):
    from open_kinematics_amd.workloads import axle_grid_problem

    program, targets = axle_grid_problem(48, 43)                # 2064 problems: 258 wavefronts of 8
    assert _assert_same(*_both_bodies(program, targets, monkeypatch)).all()
    for n in (1, 7, 9):
        assert _assert_same(*_both_bodies(program, targets[:n], monkeypatch)).all()


def test_the_reference_line_row_keeps_the_general_body(monkeypatch):
    """Programs with the reference's zero-gradient point-on-line row reject steps as a matter of course: independent solves
    stay on the general body (okx_solve_batch), the cold body exists but is not chosen - same answers either way."""
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, targets = bump_sweep_problem(256, line_mode="softnorm")
    _assert_same(*_both_bodies(program, targets, monkeypatch))


def test_a_partial_last_round_of_the_rotated_unit_loop(monkeypatch):
    """Independent solves deal a wavefront's units out rotated over the sweep (round k: unit k G + (w + 131 k) mod G).  Batches
    that fill the grid once and a part of a second round: every problem solved exactly once - the pair-mode cold body (8
    problems per wavefront) against the general body bit for bit, the lane kernel (64 per wavefront, own geometry) against the
    quad kernel to 1e-9 mm, and nothing written past the batch."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import axle_grid_problem, macpherson_grid_problem

    props = torch.cuda.get_device_properties(0)
    waves = props.multi_processor_count * 4
    program, targets = axle_grid_problem(130, 100)                       # 13000 problems
    n = min(targets.shape[0], waves * 8 + 8 * 301 + 3)                   # one full round + 301 wavefront units + a ragged one
    cold, general = _both_bodies(program, targets[:n], monkeypatch)
    assert _assert_same(cold, general).mean() > 0.9
    program, targets = macpherson_grid_problem(300, 300)                 # 90000 problems
    n = min(targets.shape[0], waves * 64 + 64 * 333 + 17)
    dp = DeviceProgram(program, "cuda:0")
    assert dp.lane_threshold > 0, dp.lane_note
    t = torch.as_tensor(targets[:n], device="cuda:0")
    guard = torch.full((n + 1, program.n_out, 3), -7.0, dtype=torch.float64, device="cuda:0")
    lane = dp.solve(t, chain_len=1, predictor=False, kernel="lane", out=guard[:n])
    quad = dp.solve(t, chain_len=1, predictor=False, kernel="quad")
    torch.cuda.synchronize()
    assert np.all((lane.info()["flags"] & 7) == 1) and np.all((quad.info()["flags"] & 7) == 1)
    assert float((lane.positions - quad.positions).abs().max()) <= 1e-9
    assert float((guard[n] + 7.0).abs().max()) == 0.0, "wrote past the batch"
    dp.close()
