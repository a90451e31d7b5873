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


def _oracle_jobs(jobs, tol=1e-9, raw_tol=3e-9, polished_share=0.01):
    """
    Every job ``(program, targets [k, T], device positions [k, n_out, 3])`` through the oracle's MINPACK at tight
    tolerances, cold start per problem, on every host core (the C oracle is re-entrant, ctypes releases the GIL).
    ``|device - oracle| <= tol`` (1e-9 mm, north_star's figure).  MINPACK itself stops up to ~1.3e-9 mm short of the
    minimiser on a fraction of a percent of the rack-steered problems (``tools/c4_oracle_gap.py``: a Gauss-Newton step on
    the oracle's OWN residuals and Jacobian moves its answer by that much and the device's by 2e-13): where the raw gap
    exceeds ``tol`` (never ``raw_tol``) the oracle's answer is polished with such steps - same objective, its fixed point -
    and the device must then agree with it to ``tol``; at most ``polished_share`` of the problems may need that.
    """
    from concurrent.futures import ThreadPoolExecutor

    from open_kinematics_amd.hostcpu import host_cores
    from oracle.oracle import Oracle

    def run(job):
        program, t, pos = job
        orc = Oracle(program)
        ref = orc.sweep(t, 1e-15, 1e-15, 1e-15, warm_start=False)
        err = np.abs(pos - ref.positions).reshape(len(t), -1).max(axis=1)
        raw, polished = float(err.max()), 0
        for k in np.nonzero(err > tol)[0]:
            x = ref.x[k].copy()
            for _ in range(3):  # Gauss-Newton on the oracle's own rows: x <- x - J^+ r
                r, jac = orc.eval(x[None], t[k][None], jac=True)
                x = x - np.linalg.lstsq(jac[0], r[0], rcond=None)[0]
            err[k] = float(np.abs(pos[k] - orc.positions(x)[program.out_point]).max())
            polished += 1
        return ref.first_failed_step, raw, float(err.max()), polished, len(t)

    with ThreadPoolExecutor(max_workers=max(1, min(32, host_cores()[0]))) as pool:
        results = list(pool.map(run, jobs))
    assert all(r[0] == -1 for r in results)
    raw, worst = max(r[1] for r in results), max(r[2] for r in results)
    n_polished, n = sum(r[3] for r in results), sum(r[4] for r in results)
    assert raw <= raw_tol, raw
    assert worst <= tol, worst
    assert n_polished <= polished_share * n, (n_polished, n)
    return worst


def _stratified(n_problems: int, n_bands: int, per_band: int, seed: int, tail: int = 8) -> np.ndarray:
    """Seeded sample of problem indices: ``per_band`` from each of ``n_bands`` equal index bands (a band = one round of the
    kernel's grid-stride loop over the chip, so every round - and with the rotated unit order every wavefront's share of it -
    is hit) plus the last ``tail`` problems (the partial last wave unit)."""
    rng = np.random.default_rng(seed)
    band = n_problems // n_bands
    pick = [rng.choice(band, size=per_band, replace=False) + b * band for b in range(n_bands)]
    pick.append(np.arange(n_problems - tail, n_problems))
    return np.unique(np.concatenate(pick))


def _oracle_sample(program, targets, pos, pick, tol=1e-9, per_job=64):
    jobs = [(program, targets[pick[i : i + per_job]], pos[pick[i : i + per_job]]) for i in range(0, len(pick), per_job)]
    return _oracle_jobs(jobs, tol)


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
    # the oracle on 4096 + seeded problems: 1024 from each of the lane kernel's four rounds over the chip, and the grid's tail
    pick = _stratified(targets.shape[0], 4, 1024, seed=4)
    assert len(pick) >= 4096
    _oracle_sample(program, targets, pos, pick)
    cold = dp.solve(t, chain_len=1, predictor=False)   # (BASELINE's rule: every problem a cold start - the same sample)
    assert cold.accepted(cold.info()).all()
    _oracle_sample(program, targets, cold.positions.cpu().numpy(), pick)
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
    # the oracle on 4096 + seeded problems: 512 from each of the pair kernel's eight rounds (8192 wave units of 8 problems on
    # 1024 wavefronts; the cold body walks them in a rotated order), and the grid's tail - chained and cold launches
    pick = _stratified(targets.shape[0], 8, 512, seed=3)
    assert len(pick) >= 4096
    _oracle_sample(program, targets, pos, pick)
    cold = dp.solve(t, chain_len=1, predictor=False)
    assert cold.accepted(cold.info()).all()
    cold_pos = cold.positions.cpu().numpy()
    _check_targets_met(program, cold_pos, targets)
    _oracle_sample(program, targets, cold_pos, pick)


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
    # ALL 1 048 576 rows, on the device: every target row met against its geometry's own targets, and four rigid links at
    # the length THEIR geometry's design state gives them (gpos: the rebound table the kernel read)
    out = [int(k) for k in program.out_point]
    P = res.positions
    for k in range(program.n_targets):
        got = P[:, out.index(int(program.tgt_point[k]))] @ torch.as_tensor(program.tgt_dir[k], device="cuda:0")
        assert float((got - targets[:, k]).abs().max()) <= 1e-9, k
    names = _names(program)
    design = gpos[:, program.out_point]                                       # [G, n_out, 3]
    for a, b in [("upper_wishbone_inboard_front", "upper_wishbone_outboard"), ("trackrod_inboard", "trackrod_outboard"),
                 ("lower_wishbone_inboard_rear", "lower_wishbone_outboard"), ("axle_inboard", "axle_outboard")]:
        ia, ib = names.index(a), names.index(b)
        length = (design[:, ia] - design[:, ib]).norm(dim=1)                   # [G]
        d = (P[:, ia] - P[:, ib]).norm(dim=1).reshape(4096, 256)
        assert float((d - length[:, None]).abs().max()) <= 5e-6, (a, b)       # (softnorm bias ~1e-6)
    # the oracle (MINPACK, tight, cold) on 4096 + seeded problems: 32 geometries from each of 16 bands of 256 (a band is one
    # round of the lane kernel's wave units over the chip; its per-geometry tables go through a rotated unit loop), 8 steps of
    # each - two from every 64-step wave unit of the geometry -, plus the last geometry's last steps; every geometry on ITS
    # own program (design positions and row parameters rebound on the CPU)
    rng = np.random.default_rng(5)
    geoms = np.unique(np.concatenate([rng.choice(256, size=32, replace=False) + 256 * b for b in range(16)] + [np.array([0, 4095])]))
    pos = res.positions.cpu().numpy().reshape(4096, 256, program.n_out, 3)
    t_host = targets.cpu().numpy().reshape(4096, 256, -1)
    base = Oracle(program)
    jobs = []
    for g in geoms:
        steps = np.concatenate([rng.choice(64, size=2, replace=False) + 64 * q for q in range(4)])
        if g == 4095:
            steps = np.unique(np.concatenate([steps, np.arange(248, 256)]))
        gp, rp = base.rebind(table[g])
        jobs.append((dataclasses.replace(program, design_pos=gp, row_param=rp), t_host[g][steps], pos[g][steps]))
    assert sum(len(job[1]) for job in jobs) >= 4096
    _oracle_jobs(jobs)
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
    # the separate launches on 64 whole geometries (seeded: four from each band of 256, first and last included), as one
    # batch of their own: tangents -> corner metrics with derivative columns
    rng = np.random.default_rng(6)
    sel = np.unique(np.concatenate([rng.choice(256, size=4, replace=False) + 256 * b for b in range(16)] + [np.array([0, 4095])]))
    assert len(sel) >= 64
    sel_t = torch.as_tensor(sel, device="cuda:0")
    rows = (sel_t[:, None] * 256 + torch.arange(256, device="cuda:0")[None]).reshape(-1)
    gk = dict(geom_pos=gpos[sel_t].contiguous(), geom_row_param=gparam[sel_t].contiguous(), steps_per_geometry=256)
    pos_sel = plain.positions[rows].contiguous()
    tan, _ = dp.tangents(pos_sel, **gk)
    sep = corner_state_metrics(roles, pos_sel, tan)
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


def test_c3_axle_grid_evaluated_at_full_size():
    """BASELINE config 3 (256 x 256 rocker + U-bar axle grid, pair mode) with every state EVALUATED in the solve's launch
    (the axle epilogue: both corners' catalogs, axle-scope metrics, rotation roles): positions and info records bit for bit the
    plain solve's, the tangent solves healthy, travel / heave identities on all 65 536 rows, a seeded sample of 2048 states
    against the six separate launches, and okx_evaluate_batch on the solved states against the fused rows."""
    from open_kinematics_amd._abi import EVAL_AXLE_METRICS
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import METRIC_NAMES, axle_evaluation_roles
    from open_kinematics_amd.workloads import axle_grid_problem, geometry_path
    from test_gpu_axle_evaluated import _assert_block_matches, _separate

    program, targets = axle_grid_problem(256, 256)
    axle = load_geometry(geometry_path("axle_geometry_rocker.yaml"))
    dp = DeviceProgram(program, "cuda:0")
    roles, rot_names, hw_names = axle_evaluation_roles(axle, program)
    dp.enable_evaluation(roles)
    assert dp.eval_columns == 64
    t = torch.as_tensor(targets, device="cuda:0")
    plain = dp.solve(t, chain_len=1, predictor=False)
    fused = dp.solve_evaluated(t, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions) and torch.equal(plain.info_raw, fused.info_raw)
    tinfo = fused.tangent_info()
    assert np.all(tinfo["flags"] == 1) and np.all(tinfo["min_pivot"] > 0.0)
    ev = fused.eval
    # the metrics' design references are each CORNER's own initial state (axle_metrics.py:29-37), microns from the axle program's
    base = torch.as_tensor([roles.left.design_wheel_center_z, roles.right.design_wheel_center_z, roles.left.design_rack_y],
                           dtype=torch.float64, device="cuda:0")
    rise = t - base[None]                                                     # [B, 3]: left hub, right hub, rack
    travel = METRIC_NAMES.index("wheel_travel")
    assert float((ev[:, 0, travel] - rise[:, 0]).abs().max()) <= 1e-8          # the hub targets ARE the wheel centres' rises
    assert float((ev[:, 0, 24 + travel] - rise[:, 1]).abs().max()) <= 1e-8
    assert float((ev[:, 0, EVAL_AXLE_METRICS + 0] - 0.5 * (rise[:, 0] + rise[:, 1])).abs().max()) <= 1e-8   # heave
    assert float((ev[:, 0, EVAL_AXLE_METRICS + 6] - rise[:, 2]).abs().max()) <= 1e-8                        # rack displacement
    # d heave / d (left hub target) = d heave / d (right hub target) = 1/2, d / d rack = 0; a hub target lifts its own wheel only
    assert float((ev[:, 1:3, EVAL_AXLE_METRICS + 0] - 0.5).abs().max()) <= 1e-9 and float(ev[:, 3, EVAL_AXLE_METRICS + 0].abs().max()) <= 1e-9
    assert float((ev[:, 1, 21] - 1.0).abs().max()) <= 1e-9 and float(ev[:, 2, 21].abs().max()) <= 1e-9
    assert float((ev[:, 2, 24 + 21] - 1.0).abs().max()) <= 1e-9 and float(ev[:, 1, 24 + 21].abs().max()) <= 1e-9
    assert bool(torch.isfinite(ev[:, :, :8]).all()) and bool(torch.isfinite(ev[:, :, 24:32]).all())
    pick = torch.as_tensor(_stratified(t.shape[0], 8, 256, seed=7), device="cuda:0")
    assert len(pick) >= 2048
    sample = dp.evaluate(plain.positions[pick].contiguous(), tangents=True)
    sep = _separate(axle, program, dp, plain.positions[pick].contiguous())
    assert set(sep["role_names"]) == set(rot_names) | set(hw_names)
    _assert_block_matches(sample, sep, roles, program)
    a, b = torch.nan_to_num(ev[pick]), torch.nan_to_num(sample.eval)
    assert float(((a - b).abs() / b.abs().clamp(min=1.0)).max()) <= 1e-9
    given = dp.evaluate(plain.positions)
    torch.cuda.synchronize()
    a, b = torch.nan_to_num(given.eval), torch.nan_to_num(ev)
    assert float(((a - b).abs() / b.abs().clamp(min=1.0)).max()) <= 1e-9
