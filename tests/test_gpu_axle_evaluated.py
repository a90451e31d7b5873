"""
GPU: the EVALUATED solve of a composed axle (okx_program_enable_axle_evaluation; the pair-mode evaluated module) - the
solve, the solution-manifold tangents, BOTH corners' metric catalogs with their derivative columns, the axle-scope
metrics and the rotation / hardware roles in ONE launch (reference core/sweep.py:113-173,217-270 for an AxleSuspension,
core/sensitivity.py:57-174, core/metrics/axle_metrics.py:47-70) - against the reference's tangent and metric goldens,
against the six separate launches and, for the positions, bit for bit against the plain solve.
"""

import os

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN, gpu_available
from test_metrics_oracle import close, load_metrics_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _axle(golden, name):
    """(suspension, sweep program as the drop-in flattens it, DeviceProgram with the axle evaluation enabled, roles ...)."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import axle_evaluation_roles
    from open_kinematics_amd.sweep import sweep_program

    arrays, _ = golden(name)
    axle = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), axle)
    program, table = sweep_program(axle, sweep)
    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad", dp.kernel_note
    roles, rot_names, hw_names = axle_evaluation_roles(axle, program)
    dp.enable_evaluation(roles)
    assert dp.evaluation & 1 and dp.eval_columns == 64, dp.evaluation_note
    return axle, sweep, program, table, dp, roles, rot_names, hw_names


def _separate(axle, program, dp, positions):
    """The six separate launches on the same states: tangents, both corners' catalogs, axle metrics, rotation and hardware roles."""
    from open_kinematics_amd.metrics import (axis_rotation_metrics, axle_roles, axle_state_metrics, corner_state_metrics,
                                             hardware_roles, topology_rotation_roles)

    tan, tinfo = dp.tangents(positions)
    left, right = axle_roles(axle, program)
    out = {"tan": tan, "tinfo": dp.tangent_info(tinfo), "left": corner_state_metrics(left, positions, tan),
           "right": corner_state_metrics(right, positions, tan), "axle": axle_state_metrics(left, right, positions)}
    names, roles = topology_rotation_roles(axle, program)
    hw_names, hw = hardware_roles(axle, program)
    if names or hw_names:
        values, rates = axis_rotation_metrics(list(roles) + list(hw), positions, tan)
        out["role_names"], out["role_values"], out["role_rates"] = list(names) + list(hw_names), values, rates
    else:
        out["role_names"] = []
    return out


def _assert_block_matches(res, sep, roles, program, value_tol=1e-9, rate_tol=1e-8):
    from open_kinematics_amd._abi import EVAL_AXLE_METRICS, EVAL_AXLE_ROLES, EVAL_RATE_WHEEL_CENTER_X
    from open_kinematics_amd.enums import PointID, PointRef, Side

    block = res.eval.cpu().numpy()
    out_keys = [program.point_keys[k] for k in program.out_point]
    tan = sep["tan"].cpu().numpy()
    for s, (tag, side) in enumerate((("left", Side.LEFT), ("right", Side.RIGHT))):
        corner = res.corner(s)
        assert close(corner.metrics.cpu().numpy(), sep[tag].values.cpu().numpy(), value_tol), tag
        assert close(corner.derivatives.cpu().numpy(), sep[tag].derivatives.cpu().numpy(), rate_tol), tag
        wc = out_keys.index(PointRef(side, PointID.WHEEL_CENTER))
        got = block[:, 1:, 24 * s + EVAL_RATE_WHEEL_CENTER_X:24 * s + EVAL_RATE_WHEEL_CENTER_X + 3]
        assert np.max(np.abs(got - tan[:, :, wc, :])) <= 1e-9, tag
    assert close(block[:, 0, EVAL_AXLE_METRICS:EVAL_AXLE_METRICS + 7], sep["axle"].cpu().numpy(), value_tol)
    for k, name in enumerate(sep["role_names"]):
        col = EVAL_AXLE_ROLES + roles.column_of[name]
        assert close(block[:, 0, col], sep["role_values"][:, k].cpu().numpy(), value_tol), name
        assert close(block[:, 1:, col], sep["role_rates"][:, :, k].cpu().numpy(), rate_tol), name
    info = res.tangent_info()
    assert np.all(info["flags"] == sep["tinfo"]["flags"])


AXLES = ["c3_axle_grid", "t_axle_t_bar_roll", "t_axle_t_bar_bump", "t_axle_heave_link"]


@pytest.mark.parametrize("name", AXLES)
def test_evaluate_given_states_matches_the_separate_launches(golden, name):
    """okx_evaluate_batch on an axle's solved states = tangents -> 2 x corner metrics -> axle metrics -> roles."""
    axle, sweep, program, table, dp, roles, rot_names, hw_names = _axle(golden, name)
    positions = dp.solve(table, chain=True).positions
    res = dp.evaluate(positions, tangents=True)
    sep = _separate(axle, program, dp, positions)
    torch.cuda.synchronize()
    assert np.max(np.abs((res.tangents - sep["tan"]).cpu().numpy())) <= 1e-9
    assert set(sep["role_names"]) == set(rot_names) | set(hw_names)
    _assert_block_matches(res, sep, roles, program)


def test_tangents_match_the_reference_goldens(golden):
    """Velocities at the reference's own solved states (tangents_c3_axle_grid.npz): <= 1e-9 mm per mm of target."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import axle_evaluation_roles
    from open_kinematics_amd.workloads import geometry_path

    from open_kinematics_amd.workloads import axle_grid_problem

    _, fixture = golden("c3_axle_grid")
    program, _ = axle_grid_problem(4, 4)  # the fixture's program with the loader's point keys (the roles are looked up by key)
    assert np.array_equal(program.out_point, fixture.out_point) and np.array_equal(program.tgt_point, fixture.tgt_point)
    tg = dict(np.load(os.path.join(GOLDEN, "tangents_c3_axle_grid.npz"), allow_pickle=False))
    axle = load_geometry(geometry_path("axle_geometry_rocker.yaml"))
    dp = DeviceProgram(program, "cuda:0")
    roles, _, _ = axle_evaluation_roles(axle, program)
    dp.enable_evaluation(roles)
    res = dp.evaluate(tg["pos"], tangents=True)
    torch.cuda.synchronize()
    assert np.all(res.tangent_info()["flags"] == 1)
    assert np.max(np.abs(res.tangents.cpu().numpy() - tg["vel"])) <= 1e-9


@pytest.mark.parametrize("chain_kw", [dict(chain_len=1), dict(chain=True), dict(chain_len=5)])
def test_evaluated_solve_positions_are_the_plain_solves_bit_for_bit(golden, chain_kw):
    """okx_solve_evaluated_batch on the C3 grid: cold body, one chain, short chains; positions and info records are the plain
    solve's bits, the evaluation block is the one okx_evaluate_batch gives on those positions."""
    from open_kinematics_amd.workloads import axle_grid_problem

    axle, sweep, _, _, _, _, _, _ = _axle(golden, "c3_axle_grid")
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.metrics import axle_evaluation_roles

    program, targets = axle_grid_problem(13, 11)   # 143 problems: a ragged last wave unit
    dp = DeviceProgram(program, "cuda:0")
    roles, _, _ = axle_evaluation_roles(axle, program)
    plain = dp.solve(targets, **chain_kw)
    res = dp.solve_evaluated(targets, roles=roles, tangents=True, **chain_kw)
    torch.cuda.synchronize()
    assert torch.equal(res.positions, plain.positions)
    assert torch.equal(res.info_raw, plain.info_raw)
    given = dp.evaluate(plain.positions, tangents=True)
    a, b = res.eval.cpu().numpy(), given.eval.cpu().numpy()
    assert close(a, b, 1e-10)
    assert np.max(np.abs((res.tangents - given.tangents).cpu().numpy())) <= 1e-11
    # metrics only: nothing but the evaluation block is written
    none = dp.solve_evaluated(targets, output="none", **chain_kw)
    assert none.positions is None and _same_bits(none.eval, res.eval)
    free = dp.solve_evaluated(targets, output="free", **chain_kw)
    assert torch.equal(dp.expand(free.free), plain.positions) and _same_bits(free.eval, res.eval)


def test_per_geometry_tables_use_each_geometrys_design_references(golden):
    """An ensemble of two geometries (the design and a copy with the left wheel centre's design height read 1 mm lower
    would be a different linkage: here the table is the design twice) gives the own-geometry block twice."""
    from open_kinematics_amd.workloads import axle_grid_problem

    axle, _, _, _, _, _, _, _ = _axle(golden, "c3_axle_grid")
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.metrics import axle_evaluation_roles

    program, targets = axle_grid_problem(3, 3)
    dp = DeviceProgram(program, "cuda:0")
    roles, _, _ = axle_evaluation_roles(axle, program)
    dp.enable_evaluation(roles)
    own = dp.solve_evaluated(targets, chain_len=1)
    hard = torch.as_tensor(np.stack([program.design_pos, program.design_pos]), device="cuda:0")
    gpos, gparam = dp.rebind(hard)
    both = dp.solve_evaluated(np.concatenate([targets, targets]), geom_pos=gpos, geom_row_param=gparam,
                              steps_per_geometry=targets.shape[0], chain_len=1)
    torch.cuda.synchronize()
    n = targets.shape[0]
    assert np.max(np.abs((both.positions[:n] - own.positions).cpu().numpy())) <= 1e-9
    assert close(both.eval[:n].cpu().numpy(), own.eval.cpu().numpy(), 1e-7)
    assert close(both.eval[n:].cpu().numpy(), own.eval.cpu().numpy(), 1e-7)


@pytest.mark.parametrize("name,metrics_name", [("c3_axle_grid", "axle_c3"), ("t_axle_heave_link", "axle_heave_link")])
def test_solve_evaluated_sweep_of_an_axle_is_one_launch_and_matches_the_reference(golden, name, metrics_name, monkeypatch):
    """The drop-in: solve_evaluated_sweep(AxleSuspension) = solve_sweep + compute_sweep_metrics, from ONE kernel launch
    (fused=True, and the default: a warm-started sweep is first solved as cold starts side by side) or from the solve and one
    evaluation launch on its records (fused=False; what the sequential chain takes when the cold starts are not kept)."""
    from open_kinematics_amd import batch, sweep as sweep_mod
    from open_kinematics_amd.enums import Side
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import AXLE_METRIC_NAMES

    arrays, _ = golden(name)
    axle = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), axle)
    calls = []
    for fn in ("tangents", "evaluate", "solve"):
        original = getattr(batch.DeviceProgram, fn)
        monkeypatch.setattr(batch.DeviceProgram, fn, lambda self, *a, _o=original, _n=fn, **k: (calls.append(_n), _o(self, *a, **k))[1])
    evaluated = sweep_mod.solve_evaluated_sweep(axle, sweep, fused=True)
    assert calls == []  # neither a plain solve nor a separate evaluation: the fused launch only
    default = sweep_mod.solve_evaluated_sweep(axle, sweep)
    assert calls == [] and len(default.states) == len(evaluated.states)
    two = sweep_mod.solve_evaluated_sweep(axle, sweep, fused=False)  # the solve, then ONE launch on its records
    assert calls == ["solve", "evaluate"]
    for row, ref in zip(two.metrics.rows, evaluated.metrics.rows):
        assert all(_same(row.axle[n], ref.axle[n]) for n in ref.axle)
        assert all(_same(row.corners[side][n], ref.corners[side][n]) for side in ref.corners for n in ref.corners[side])
    calls.clear()
    states, stats = sweep_mod.solve_sweep(axle, sweep)
    separate = sweep_mod.compute_sweep_metrics(axle, sweep, states)
    assert len(evaluated.states) == len(states) == len(evaluated.metrics.rows)
    for a, b in zip(evaluated.states, states):
        for key in a.positions:
            assert np.array_equal(np.asarray(a.positions[key].data), np.asarray(b.positions[key].data))
    # (against the reference's goldens at the reference's own states: tests/test_gpu_sweep_metrics.py, which takes the same
    #  evaluated path through compute_sweep_metrics; here the solved states are this solver's own)
    assert load_metrics_golden(metrics_name)["axle_values"].shape[1] == len(AXLE_METRIC_NAMES)
    for s, (row, ref) in enumerate(zip(evaluated.metrics.rows, separate.rows)):
        assert list(row.axle) == list(ref.axle)
        for n in row.axle:
            assert _same(row.axle[n], ref.axle[n]), (s, n)
        for side in (Side.LEFT, Side.RIGHT):
            assert list(row.corners[side]) == list(ref.corners[side])
            for n in row.corners[side]:
                assert _same(row.corners[side][n], ref.corners[side][n]), (s, side, n)


def _same_bits(a, b):
    """Equal bit for bit (NaN = the reference's None compares equal to itself)."""
    return torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64))


def _same(a, b, tol=1e-7):
    if a is None or b is None:
        return a is None and b is None
    return abs(a - b) <= tol * max(1.0, abs(b))
