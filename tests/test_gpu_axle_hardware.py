"""
GPU: the axle's shared hardware through the drop-in - rigid T-bar anti-roll bar (60 variables; pair-mode quad kernel with
three joining rows: rack, crossbar length, crossbar midpoint on the centre plane), a rocker-to-rocker heave link on the
U-bar axle and on the T-bar axle (66 variables: pair mode with 11 free points per half).  The 66-variable programs also
run through the interpreter's two-wavefront instantiation (LDL^T rows in LDS), forced here, so that the capacity path
for programs without a generated kernel stays pinned.  After the reference's tests/test_t_bar_arb.py:96-196 and tests/test_axle_rocker.py:143-175, plus the parity of the
two-wavefront kernels against the oracle and against the one-wavefront kernels on a program both can run.
"""

import copy
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN, gpu_available

pytestmark = pytest.mark.gpu
GEOM = os.path.join(GOLDEN, "geometry")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _t_bar():
    from open_kinematics_amd.input import load_geometry

    return load_geometry(os.path.join(GEOM, "axle_geometry_t_bar.yaml"))


def test_bump_sweep_preserves_rigid_t_bar_distances():
    """Same-direction wheel travel moves the T-bar stem through its XZ arc (test_t_bar_arb.py:96-137)."""
    from open_kinematics_amd.enums import PointID, PointRef, Side
    from open_kinematics_amd.input import load_sweep
    from open_kinematics_amd.sweep import solve_sweep

    axle = _t_bar()
    sweep = load_sweep(os.path.join(GEOM, "axle_t_bar_bump_sweep.yaml"), axle)
    states, stats = solve_sweep(axle, sweep)
    assert all(info.converged for info in stats)
    pivot, wheel = PointRef(Side.CENTER, PointID.ARB_T_BAR_PIVOT), PointRef(Side.LEFT, PointID.WHEEL_CENTER)
    left, right = PointRef(Side.LEFT, PointID.DROPLINK_T_BAR), PointRef(Side.RIGHT, PointID.DROPLINK_T_BAR)
    design = axle.initial_state().positions
    lengths = {pair: float(np.linalg.norm(design[pair[0]].data - design[pair[1]].data))
               for pair in ((left, right), (left, pivot), (right, pivot))}
    center_x, travel = [], []
    for state in states:
        p = state.positions
        center = p[left].data + (p[right].data - p[left].data) / 2.0
        assert abs(center[1]) <= 1e-7
        for (a, b), length in lengths.items():
            assert abs(float(np.linalg.norm(p[a].data - p[b].data)) - length) <= 1e-5
        center_x.append(float(center[0]))
        travel.append(float(p[wheel].data[2] - design[wheel].data[2]))
    assert min(travel) == pytest.approx(-50.0, abs=1e-5) and max(travel) == pytest.approx(50.0, abs=1e-5)
    assert max(center_x) - min(center_x) > 3.0


def test_roll_sweep_produces_differential_t_bar_twist():
    """Opposed wheel travel rotates the rigid crossbar about the T-bar stem (test_t_bar_arb.py:140-196)."""
    from open_kinematics_amd.input import load_sweep
    from open_kinematics_amd.sweep import AxleMetricRows, compute_sweep_metrics, solve_sweep

    axle = _t_bar()
    sweep = load_sweep(os.path.join(GEOM, "axle_t_bar_roll_sweep.yaml"), axle)
    states, stats = solve_sweep(axle, sweep)
    assert all(info.converged for info in stats)
    result = compute_sweep_metrics(axle, sweep, states)
    assert result.derivative_error is None
    expected = {"deriv_t_bar_center_x_wrt_hub_z_left", "deriv_t_bar_center_x_wrt_hub_z_right",
                "deriv_arb_twist_wrt_hub_z_left", "deriv_arb_twist_wrt_hub_z_right"}
    twists = []
    for row in result.rows:
        assert isinstance(row, AxleMetricRows)
        assert expected <= row.axle.keys() and all(row.axle[k] is not None for k in expected)
        assert "t_bar_twist" not in row.axle
        twists.append(float(row.axle["arb_twist"]))
    assert max(twists) - min(twists) > 1.0


def _heave_axle():
    from open_kinematics_amd.input import build_suspension

    with open(os.path.join(GEOM, "axle_geometry_rocker.yaml"), encoding="utf-8") as fh:
        data = copy.deepcopy(yaml.safe_load(fh))
    data["axle_config"]["heave_link"] = {"type": "rocker_to_rocker"}
    data["hardpoints"]["left"]["heave_link_rocker"] = {"x": 0, "y": 300.0, "z": 400}
    return build_suspension(data)


def test_heave_link_length_follows_the_rockers(golden):
    """The link is free to change length: no row holds it, the metric reports it (test_axle_rocker.py:143-175);
    positions against the reference's solve of the same sweep."""
    from open_kinematics_amd.enums import PointID, PointRef, Side
    from open_kinematics_amd.input import load_sweep
    from open_kinematics_amd.sweep import compute_sweep_metrics, solve_sweep

    arrays, program = golden("t_axle_heave_link")
    axle = _heave_axle()
    sweep = load_sweep(os.path.join(GEOM, "axle_rocker_sweep.yaml"), axle)
    states, stats = solve_sweep(axle, sweep)
    assert all(info.converged for info in stats)
    out = axle.output_points()
    pos = np.asarray([[s.positions[k].data for k in out] for s in states])
    assert np.max(np.abs(pos - arrays["ref_tight_pos"])) <= 1e-7    # the drop-in solves to the device tolerances
    assert np.max(np.abs(pos - arrays["ref_default_pos"])) <= 5e-5  # where the reference's default tolerances stop it
    ends = PointRef(Side.LEFT, PointID.HEAVE_LINK_ROCKER), PointRef(Side.RIGHT, PointID.HEAVE_LINK_ROCKER)
    result = compute_sweep_metrics(axle, sweep, states)
    assert result.derivative_error is None
    lengths = []
    for state, row in zip(states, result.rows):
        length = float(np.linalg.norm(state.positions[ends[0]].data - state.positions[ends[1]].data))
        assert row.axle["heave_link_length"] == pytest.approx(length, abs=1e-9)
        assert row.axle["deriv_heave_link_length_wrt_hub_z_left"] is not None
        lengths.append(length)
    assert max(lengths) - min(lengths) > 1e-3


@pytest.mark.parametrize("name", ["t_axle_t_bar_heave", "t_axle_heave_link"])
def test_two_wavefront_kernels_match_the_oracle(golden, name, monkeypatch):
    """R1 / R1b / solve for a program of more than 63 variables: residuals, Jacobian, J^T J, J^T r at seeded points and
    the solved sweep, cold and chained, against the oracle; tangents against a central difference of solves."""
    from open_kinematics_amd.batch import DeviceProgram
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    assert pinned.n_vars > 63
    dp = DeviceProgram(pinned, "cuda:0")
    # both have a generated pair-mode kernel (one and three joining rows); the interpreter instantiation is forced here
    assert dp.kernel == "quad", dp.kernel_note
    force = {"kernel": "single"}
    monkeypatch.setenv("OKX_DEV", "tangent_generic")  # okx_tangent_batch: the interpreter's tangent kernel
    x, t = arrays["eval_x"], arrays["eval_targets"]
    r_o, jac_o = Oracle(pinned).eval(x, t)
    r, jac = dp.eval(x, t)
    assert np.all(np.abs(r.cpu().numpy() - r_o) <= 2.5e-13 + 1e-13 * np.abs(r_o))
    assert np.max(np.abs(jac.cpu().numpy() - jac_o)) <= 1e-12 * max(1.0, np.abs(jac_o).max())
    _, ata, atr = dp.normal_equations(x, t)
    ata_o, atr_o = np.einsum("bij,bik->bjk", jac_o, jac_o), np.einsum("bij,bi->bj", jac_o, r_o)
    assert np.max(np.abs(ata.cpu().numpy() - ata_o)) <= 1e-11 * max(1.0, np.abs(ata_o).max())
    assert np.max(np.abs(atr.cpu().numpy() - atr_o)) <= 1e-11 * max(1.0, np.abs(atr_o).max())
    targets = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    orc = Oracle(pinned).sweep(arrays["targets_abs"], 1e-15, 1e-15, 1e-15, warm_start=False)
    for kw in (dict(chain_len=1), dict(chain=True)):
        res = dp.solve(targets, **kw, **force)
        torch.cuda.synchronize()
        info = res.info()
        assert np.all((info["flags"] & 7) == 1), info["flags"]
        pos = res.positions.cpu().numpy()
        assert np.max(np.abs(pos - orc.positions)) <= 1e-9                # north-star tolerance (mm)
        assert np.max(np.abs(pos - arrays["ref_tight_pos"])) <= 6e-8     # the reference's own floor
    # the generated pair-mode kernel against the interpreter on the same program
    quad = dp.solve(targets, chain_len=1)
    assert np.all((quad.info()["flags"] & 7) == 1)
    assert float((quad.positions - dp.solve(targets, chain_len=1, **force).positions).abs().max()) <= 1e-10
    assert np.max(np.abs(quad.positions.cpu().numpy() - orc.positions)) <= 1e-9
    # tangents (the interpreter's kernel, forced above; then the generated one): d positions / d target against central
    # differences of tight solves
    res = dp.solve(targets, chain_len=1)
    tan, tinfo = dp.tangents(res.positions)
    monkeypatch.delenv("OKX_DEV")
    tan_quad, tinfo_quad = dp.tangents(res.positions)
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo_quad)["flags"] == 1)
    assert float((tan - tan_quad).abs().max()) <= 1e-9 * max(1.0, float(tan.abs().max()))
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    h = 1e-3
    for k in range(targets.shape[1]):
        d = torch.zeros_like(targets)
        d[:, k] = h
        fd = (dp.solve(targets + d, chain_len=1).positions - dp.solve(targets - d, chain_len=1).positions) / (2 * h)
        assert float((tan[:, k] - fd).abs().max()) <= 2e-6 * max(1.0, float(fd.abs().max()))


def test_two_wavefront_factorisation_gives_the_damped_step(golden):
    """The LDL^T of the interpreter's two-wavefront kernels (rows in LDS, right-hand side as row n): the first damped
    step of a 66-variable program against numpy on the kernel's own J^T J."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("t_axle_t_bar_heave")
    pinned = program.with_line_mode("pinned")
    dp = DeviceProgram(pinned, "cuda:0")
    assert pinned.n_vars == 66
    targets = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    # one LM step from the design state with max_iter = 1 is x0 - (J^T J + lambda I)^-1 J^T r
    res = dp.solve(targets, chain_len=1, max_iter=1, shared_first_step=False, kernel="single")
    torch.cuda.synchronize()
    n = pinned.n_vars
    x0 = pinned.design_pos[pinned.free_point].reshape(-1)
    xs = np.repeat(x0[None], targets.shape[0], 0)
    _, ata, atr = [v.cpu().numpy() for v in dp.normal_equations(xs, arrays["targets_abs"])]
    free_out = [list(pinned.out_point).index(p) for p in pinned.free_point]
    got = res.positions.cpu().numpy()[:, free_out].reshape(-1, n) - xs
    for b in range(targets.shape[0]):
        lam = dp.default_opts().lambda0 * np.max(np.diag(ata[b]))  # the damping a cold start begins with
        want = -np.linalg.solve(ata[b] + lam * np.eye(n), atr[b])
        assert np.max(np.abs(got[b] - want)) <= 1e-9 * max(1.0, np.abs(want).max()), b


@pytest.mark.parametrize("name", ["t_axle_t_bar_roll", "t_axle_t_bar_bump", "t_axle_t_bar_heave"])
def test_pair_mode_with_three_joining_rows_takes_the_coupled_step(golden, name):
    """The T-bar axle in pair mode: rack length, crossbar length and crossbar midpoint plane join the halves (a 6 x 6
    Woodbury system on per-half dot products).  Its first damped step from the design state against numpy on the oracle's
    Jacobian, then cold and chained solves against the interpreter and the reference."""
    from open_kinematics_amd.batch import DeviceProgram
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    dp = DeviceProgram(pinned, "cuda:0")
    assert dp.kernel == "quad", dp.kernel_note
    n = pinned.n_vars
    t_host = arrays["targets_abs"].reshape(-1, pinned.n_targets)
    targets = torch.as_tensor(t_host, device="cuda:0")
    x0 = pinned.design_pos[pinned.free_point].reshape(-1)
    xs = np.repeat(x0[None], len(t_host), 0)
    r_o, jac_o = Oracle(pinned).eval(xs, t_host)
    ata, atr = np.einsum("bij,bik->bjk", jac_o, jac_o), np.einsum("bij,bi->bj", jac_o, r_o)
    res = dp.solve(targets, chain_len=1, max_iter=1, shared_first_step=False, predictor=False)
    torch.cuda.synchronize()
    free_out = [list(pinned.out_point).index(p) for p in pinned.free_point]
    got = res.positions.cpu().numpy()[:, free_out].reshape(-1, n) - xs
    lam0 = dp.default_opts().lambda0
    accepted = 0
    for b in range(len(t_host)):
        want = -np.linalg.solve(ata[b] + lam0 * np.max(np.diag(ata[b])) * np.eye(n), atr[b])
        if np.abs(got[b]).max() > 0.0:  # (a rejected trial leaves the design state in place)
            accepted += 1
            assert np.max(np.abs(got[b] - want)) <= 1e-9 * max(1.0, np.abs(want).max()), b
    assert accepted >= len(t_host) // 2
    wave = dp.solve(targets, chain_len=1, kernel="single").positions
    for kw in (dict(chain_len=1), dict(chain=True), dict(chain_len=1, shared_first_step=False)):
        out = dp.solve(targets, **kw)
        assert np.all((out.info()["flags"] & 7) == 1)
        assert float((out.positions - wave).abs().max()) <= 1e-10
        assert np.max(np.abs(out.positions.cpu().numpy() - arrays["ref_tight_pos"])) <= 6e-8


@pytest.mark.parametrize("name", ["t_axle_t_bar_roll", "t_axle_heave_link", "t_axle_t_bar_heave", "t_corner_strut_rocker"])
def test_reference_literal_line_rows_on_the_generated_kernels(golden, name):
    """line_mode = softnorm (the reference's zero-gradient point-on-line rows, as flattened from its objects) on the
    kernels round 3 added: three joining rows, 11 free points per half, the nine-point corner.  Linear convergence along
    the valley (looser step tolerance, more passes), inside the reference's own floor."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden(name)
    assert program.line_mode == "softnorm"
    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad", dp.kernel_note
    targets = torch.as_tensor(arrays["targets_abs"].reshape(-1, program.n_targets), device="cuda:0")
    res = dp.solve(targets, chain_len=1, step_tol=1e-8, max_iter=200)
    torch.cuda.synchronize()
    info = res.info()
    assert res.accepted(info).all()
    ref = arrays["ref_tight_pos"].reshape(len(targets), -1, 3)
    assert np.max(np.abs(res.positions.cpu().numpy() - ref)) <= 6e-8


def _stack(a, b):
    """Two independent constraint programs as one: b's points, rows, derived ops, targets and outputs after a's."""
    from open_kinematics_amd.program import ConstraintProgram, NamedKey

    pa = a.n_points

    def shift(pts):
        pts = np.asarray(pts).copy()
        pts[pts >= 0] += pa
        return pts

    return ConstraintProgram(
        point_keys=[NamedKey("a_" + str(k)) for k in range(pa)] + [NamedKey("b_" + str(k)) for k in range(b.n_points)],
        role=np.concatenate([a.role, b.role]),
        free_point=np.concatenate([a.free_point, shift(b.free_point)]).astype(np.int32),
        dop_type=np.concatenate([a.dop_type, b.dop_type]).astype(np.int32),
        dop_out=np.concatenate([a.dop_out, shift(b.dop_out)]).astype(np.int32),
        dop_pts=np.concatenate([a.dop_pts.reshape(-1, 4), shift(b.dop_pts.reshape(-1, 4))]).astype(np.int32),
        dop_param=np.concatenate([a.dop_param, b.dop_param]),
        row_type=np.concatenate([a.row_type, b.row_type]).astype(np.int32),
        row_pts=np.concatenate([a.row_pts, shift(b.row_pts)]).astype(np.int32),
        row_param=np.concatenate([a.row_param, b.row_param]),
        row_source=np.concatenate([a.row_source, b.row_source + (a.row_source.max() + 1 if len(a.row_source) else 0)]).astype(np.int32),
        tgt_point=np.concatenate([a.tgt_point, shift(b.tgt_point)]).astype(np.int32),
        tgt_dir=np.concatenate([a.tgt_dir, b.tgt_dir]),
        out_point=np.concatenate([a.out_point, shift(b.out_point)]).astype(np.int32),
        design_pos=np.concatenate([a.design_pos, b.design_pos]),
        constraint_desc=list(a.constraint_desc) + list(b.constraint_desc),
        target_desc=list(a.target_desc) + list(b.target_desc),
        line_mode=a.line_mode,
    )


def test_two_wavefront_interpreter_at_ninety_six_variables(golden):
    """Capacity beyond the reference's own compositions: the rocker + U-bar axle and the plain double-wishbone axle as
    ONE program (96 variables, no pair structure, so no generated kernel).  The interpreter's two-wavefront kernels on it
    against each part solved alone and against the oracle."""
    from open_kinematics_amd.batch import DeviceProgram
    from oracle.oracle import Oracle

    arr_a, prog_a = golden("c3_axle_grid")
    arr_b, prog_b = golden("t_axle_dw")
    a, b = prog_a.with_line_mode("pinned"), prog_b.with_line_mode("pinned")
    both = _stack(a, b)
    both.validate()
    assert both.n_vars == a.n_vars + b.n_vars == 96
    n = min(len(arr_a["targets_abs"]), len(arr_b["targets_abs"]), 9)
    t_a, t_b = arr_a["targets_abs"][:n], arr_b["targets_abs"][:n]
    targets = np.concatenate([t_a, t_b], axis=1)
    dp = DeviceProgram(both, "cuda:0")
    assert dp.kernel == "wave", "a stacked program has no pair structure"
    x = np.concatenate([arr_a["eval_x"][:4], arr_b["eval_x"][:4]], axis=1)
    t_eval = np.concatenate([arr_a["eval_targets"][:4], arr_b["eval_targets"][:4]], axis=1)
    r_o, jac_o = Oracle(both).eval(x, t_eval)
    r, jac = dp.eval(x, t_eval)
    assert np.all(np.abs(r.cpu().numpy() - r_o) <= 2.5e-13 + 1e-13 * np.abs(r_o))
    assert np.max(np.abs(jac.cpu().numpy() - jac_o)) <= 1e-12 * max(1.0, np.abs(jac_o).max())
    res = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain_len=1)
    torch.cuda.synchronize()
    assert np.all((res.info()["flags"] & 7) == 1)
    pos = res.positions.cpu().numpy()
    alone_a = DeviceProgram(a, "cuda:0").solve(torch.as_tensor(t_a, device="cuda:0"), chain_len=1, kernel="single").positions.cpu().numpy()
    alone_b = DeviceProgram(b, "cuda:0").solve(torch.as_tensor(t_b, device="cuda:0"), chain_len=1, kernel="single").positions.cpu().numpy()
    assert np.max(np.abs(pos[:, : a.n_out] - alone_a)) <= 1e-9 and np.max(np.abs(pos[:, a.n_out:] - alone_b)) <= 1e-9
    orc = Oracle(both).sweep(targets[:3], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert np.max(np.abs(pos[:3] - orc.positions)) <= 1e-9
