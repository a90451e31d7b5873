"""
GPU: BASELINE configs 3, 4 and 5 at their FULL sizes (256 x 256 axle grid, 512 x 512 MacPherson grid,
4096 geometries x 256 steps) through size-independent properties, plus the oracle on a sample of each.
(Config 2 at full size: tests/test_gpu_parity.py::test_full_size_bump_sweep_properties.)
"""

import dataclasses

import numpy as np
import pytest
import torch

from conftest import gpu_available

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _names(program):
    from open_kinematics_amd.results_writer import point_key_name

    return [point_key_name(program.point_keys[k]) for k in program.out_point]


def _check_targets_met(program, pos, targets, tol=1e-9):
    """Every target row: dir . p(point) == target (absolute)."""
    out = list(program.out_point)
    for k in range(program.n_targets):
        idx = out.index(int(program.tgt_point[k]))
        got = pos[:, idx] @ np.asarray(program.tgt_dir[k])
        assert np.max(np.abs(got - targets[:, k])) <= tol, k


def _lengths_preserved(program, pos, pairs, tol=5e-6):
    names = _names(program)
    design = program.design_pos[program.out_point]
    for a, b in pairs:
        ia, ib = names.index(a), names.index(b)
        d = np.linalg.norm(pos[:, ia] - pos[:, ib], axis=1)
        assert np.max(np.abs(d - np.linalg.norm(design[ia] - design[ib]))) <= tol, (a, b)  # softnorm bias ~1e-6


def _oracle_sample(program, targets, pos, n=24, tol=1e-9):
    from oracle.oracle import Oracle

    pick = np.linspace(0, targets.shape[0] - 1, n).astype(int)
    orc = Oracle(program).sweep(targets[pick], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert orc.first_failed_step == -1
    assert np.max(np.abs(pos[pick] - orc.positions)) <= tol


def test_c4_macpherson_512x512_grid():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import macpherson_grid_problem

    program, targets = macpherson_grid_problem(512, 512)
    dp = DeviceProgram(program, "cuda:0")
    t = torch.as_tensor(targets, device="cuda:0")
    res = dp.solve(t, chain_len=-1)
    info = res.info()
    assert res.accepted(info).all() and info["max_residual"].max() <= 2e-6
    pos = res.positions.cpu().numpy()
    _check_targets_met(program, pos, targets)
    _lengths_preserved(program, pos, [("lower_wishbone_inboard_front", "lower_wishbone_outboard"),
                                      ("trackrod_inboard", "trackrod_outboard"), ("axle_inboard", "axle_outboard")])
    # start point independence: a strided subset solved cold and alone gives the same states
    pick = torch.arange(0, t.shape[0], 257, device="cuda:0")
    alone = dp.solve(t[pick].contiguous(), chain_len=1, predictor=False).positions
    assert float((alone - res.positions[pick]).abs().max()) <= 1e-9
    _oracle_sample(program, targets, pos)
    # quad kernel: chains of 16 (one per resident quad), about one full pass and a confirmation per solve.  The lane kernel
    # (auto selection beyond one round of the quad kernel) holds four times as many problems at once, so its chains are 4 steps long and
    # the cold head weighs more - and it is still the faster launch (bench.py other_configs).
    assert info["nfev"].mean() <= (3.4 if dp.lane_bodies & 2 else 2.6)
    quad = dp.solve(t, chain_len=-1, kernel="quad")
    assert quad.info()["nfev"].mean() <= 2.6
    assert float((quad.positions - res.positions).abs().max()) <= 1e-9


def test_c3_rocker_axle_256x256_grid():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import axle_grid_problem

    program, targets = axle_grid_problem(256, 256)
    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad"  # pair mode: one quad per corner
    t = torch.as_tensor(targets, device="cuda:0")
    res = dp.solve(t, chain_len=-1)
    info = res.info()
    assert res.accepted(info).all() and info["max_residual"].max() <= 2e-6
    pos = res.positions.cpu().numpy()
    _check_targets_met(program, pos, targets)
    _lengths_preserved(program, pos, [("left_trackrod_inboard", "right_trackrod_inboard"),   # the rigid rack
                                      ("left_pushrod_outboard", "left_pushrod_inboard"),
                                      ("right_droplink_rocker", "right_droplink_u_bar"),
                                      ("left_upper_wishbone_inboard_rear", "left_upper_wishbone_outboard")])
    pick = torch.arange(0, t.shape[0], 1031, device="cuda:0")
    alone = dp.solve(t[pick].contiguous(), chain_len=1).positions
    assert float((alone - res.positions[pick]).abs().max()) <= 1e-9
    _oracle_sample(program, targets, pos, n=12)


def test_c5_ensemble_4096_geometries_x_256_steps():
    from oracle.oracle import Oracle
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import ensemble_problem

    program, table, rel = ensemble_problem(4096, 256)
    dp = DeviceProgram(program, "cuda:0")
    gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda:0"))
    base = torch.stack([gpos[:, program.tgt_point[k]] @ torch.as_tensor(program.tgt_dir[k], device="cuda:0")
                        for k in range(program.n_targets)], 1)
    targets = (base[:, None, :] + torch.as_tensor(rel, device="cuda:0")[None]).reshape(-1, program.n_targets).contiguous()
    res = dp.solve(targets, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=256, chain_len=-1)
    info = res.info()
    assert res.accepted(info).all() and info["max_residual"].max() <= 2e-6
    pos = res.positions.cpu().numpy().reshape(4096, 256, program.n_out, 3)
    t_host = targets.cpu().numpy().reshape(4096, 256, -1)
    for g in (0, 1777, 4095):  # three geometries in full: targets met, links keep THEIR geometry's lengths, oracle on a sample
        _check_targets_met(program, pos[g], t_host[g])
        names = _names(program)
        design = gpos[g].cpu().numpy()[program.out_point]
        for a, b in [("upper_wishbone_inboard_front", "upper_wishbone_outboard"), ("trackrod_inboard", "trackrod_outboard")]:
            ia, ib = names.index(a), names.index(b)
            d = np.linalg.norm(pos[g][:, ia] - pos[g][:, ib], axis=1)
            assert np.max(np.abs(d - np.linalg.norm(design[ia] - design[ib]))) <= 5e-6
        # the oracle on this geometry's own program (design positions and row parameters rebound on the CPU)
        gp, rp = Oracle(program).rebind(table[g])
        bound = Oracle(dataclasses.replace(program, design_pos=gp, row_param=rp))
        pick = np.linspace(0, 255, 8).astype(int)
        ref = bound.sweep(t_host[g][pick], 1e-15, 1e-15, 1e-15, warm_start=False)
        assert ref.first_failed_step == -1 and np.max(np.abs(pos[g][pick] - ref.positions)) <= 1e-9
    # geometry-major independence: geometry 1777 solved alone gives the same states
    alone = dp.solve(targets.reshape(4096, 256, -1)[1777].contiguous(), geom_pos=gpos[1777:1778].contiguous(),
                     geom_row_param=gparam[1777:1778].contiguous(), steps_per_geometry=256, chain_len=1).positions
    assert float((alone - res.positions.reshape(4096, 256, -1, 3)[1777]).abs().max()) <= 1e-9


def test_c5_ensemble_evaluated_at_full_size():
    """BASELINE config 5, every state EVALUATED in the solve's launch (okx_solve_evaluated_batch) - size-independent
    properties of the 1 048 576 evaluation rows, the separate launches on a sample, and okx_evaluate_batch (lane form) on
    the solved states against the fused rows."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_roles, corner_state_metrics
    from open_kinematics_amd.workloads import ensemble_problem, geometry_path

    program, table, rel = ensemble_problem(4096, 256)
    dp = DeviceProgram(program, "cuda:0")
    roles = corner_roles(load_geometry(geometry_path("geometry.yaml")), program)
    dp.enable_evaluation(roles)
    assert dp.evaluation == 3, dp.evaluation_note
    gpos, gparam = dp.rebind(torch.as_tensor(table, device="cuda:0"))
    targets = dp.ensemble_targets(gpos, rel)
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=256)
    plain = dp.solve(targets, chain_len=1, **kw)
    fused = dp.solve_evaluated(targets, chain_len=1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions) and torch.equal(plain.info_raw, fused.info_raw)
    info = fused.tangent_info()
    assert np.all(info["flags"] == 1) and np.all(info["min_pivot"] > 0.0)
    ev = fused.eval
    bump = program.n_targets - 1
    travel = ev[:, 0, METRIC_NAMES.index("wheel_travel")].reshape(4096, 256)
    want = torch.as_tensor(rel[:, bump], device="cuda:0")[None].expand(4096, -1)
    assert float((travel - want).abs().max()) <= 1e-8                    # the bump target IS the wheel centre's rise
    # wheel travel = z - z0: its derivative along a target is the wheel centre's z rate, and along the bump target that is 1
    d_travel = ev[:, 1:, METRIC_NAMES.index("wheel_travel")]
    assert float((d_travel - ev[:, 1:, 21]).abs().max()) <= 1e-12
    assert float((ev[:, 1 + bump, 21] - 1.0).abs().max()) <= 1e-9
    assert float(ev[:, 1, 21].abs().max()) <= 1e-9                     # ... and the rack target does not lift the wheel centre
    assert bool(torch.isfinite(ev[:, :, :8]).all())                      # angles, travel, track, scrub, trail: defined everywhere here
    # the separate launches on four whole geometries
    for g in (0, 1234, 2048, 4095):
        rows = slice(g * 256, (g + 1) * 256)
        gk = dict(geom_pos=gpos[g : g + 1], geom_row_param=gparam[g : g + 1], steps_per_geometry=256)
        tan, _ = dp.tangents(plain.positions[rows], **gk)
        sep = corner_state_metrics(roles, plain.positions[rows], tan)
        cols = [k for k, n in enumerate(METRIC_NAMES) if n != "wheel_travel"]   # (that one is measured from the geometry's own design state)
        a, b = ev[rows, 0][:, cols].cpu().numpy(), sep.values[:, cols].cpu().numpy()
        both = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), both)   # (instant centres lie up to 1e4 mm away: the bound is relative there)
        assert np.max(np.abs(a[both] - b[both]) / np.maximum(1.0, np.abs(b[both]))) <= 1e-9
        da, db = ev[rows, 1:, :19].cpu().numpy(), sep.derivatives.cpu().numpy()
        both = np.isfinite(db)
        assert np.max(np.abs(da[both] - db[both]) / np.maximum(1.0, np.abs(db[both]))) <= 1e-7
    # the same epilogue on the given states, lane form (a batch that fills the chip takes it by itself)
    given = dp.evaluate(plain.positions, **kw)
    torch.cuda.synchronize()
    a, b = torch.nan_to_num(given.eval), torch.nan_to_num(ev)
    assert float(((a - b).abs() / b.abs().clamp(min=1.0)).max()) <= 1e-9


@pytest.mark.parametrize("which", ["c4", "c2"])
def test_own_geometry_configs_evaluated_at_full_size(which):
    """BASELINE config 4 (MacPherson 512 x 512, lane kernels) and config 2 (one 16384-step sweep, quad cold body) with
    every state evaluated in the solve's launch: positions bit for bit the plain solve's, the tangent solves healthy, the
    wheel-travel identities, and a strided sample against the separate launches."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_roles, corner_state_metrics
    from open_kinematics_amd.workloads import bump_sweep_problem, geometry_path, macpherson_grid_problem

    if which == "c4":
        program, targets = macpherson_grid_problem(512, 512)
        yaml_name = "macpherson_geometry.yaml"
    else:
        program, targets = bump_sweep_problem(16384)
        yaml_name = "geometry.yaml"
    dp = DeviceProgram(program, "cuda:0")
    roles = corner_roles(load_geometry(geometry_path(yaml_name)), program)
    dp.enable_evaluation(roles)
    t = torch.as_tensor(targets, device="cuda:0")
    plain = dp.solve(t, chain_len=1, predictor=False)
    fused = dp.solve_evaluated(t, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions) and torch.equal(plain.info_raw, fused.info_raw)
    assert np.all(fused.tangent_info()["flags"] == 1)
    ev = fused.eval
    bump = program.n_targets - 1
    wc = list(program.out_point).index(int(program.tgt_point[bump]))
    z0 = float(program.design_pos[program.tgt_point[bump]][2])
    travel = ev[:, 0, METRIC_NAMES.index("wheel_travel")]
    assert float((travel - (plain.positions[:, wc, 2] - z0)).abs().max()) <= 1e-12
    assert float((travel - (t[:, bump] - z0)).abs().max()) <= 1e-8
    assert float((ev[:, 1:, METRIC_NAMES.index("wheel_travel")] - ev[:, 1:, 21]).abs().max()) <= 1e-12
    assert float((ev[:, 1 + bump, 21] - 1.0).abs().max()) <= 1e-9
    pick = torch.arange(0, t.shape[0], 257 if which == "c4" else 31, device="cuda:0")
    tan, _ = dp.tangents(plain.positions[pick].contiguous())
    sep = corner_state_metrics(roles, plain.positions[pick].contiguous(), tan)
    a, b = ev[pick, 0, :19].cpu().numpy(), sep.values.cpu().numpy()
    both = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), both)
    assert np.max(np.abs(a[both] - b[both]) / np.maximum(1.0, np.abs(b[both]))) <= 1e-9
    da, db = ev[pick, 1:, :19].cpu().numpy(), sep.derivatives.cpu().numpy()
    both = np.isfinite(db)
    assert np.max(np.abs(da[both] - db[both]) / np.maximum(1.0, np.abs(db[both]))) <= 1e-7
    given = dp.evaluate(plain.positions)   # (c4: the lane form; c2: 16384 states stay on the quad form)
    torch.cuda.synchronize()
    a, b = torch.nan_to_num(given.eval), torch.nan_to_num(ev)
    assert float(((a - b).abs() / b.abs().clamp(min=1.0)).max()) <= 1e-9
