"""
GPU: solution-manifold tangents from the generated tangent kernel, through the
C-ABI (okx_tangent_batch), against the reference's compute_state_tangents outputs, the numpy
oracle and finite differences of device solves.
"""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, gpu_available

pytestmark = pytest.mark.gpu

FIXTURES = ["c1_dw_corner", "c4_macpherson_grid", "u_dw_corner", "u_macpherson"]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _tg(name):
    return dict(np.load(os.path.join(GOLDEN, f"tangents_{name}.npz"), allow_pickle=False))


def _dp(program):
    from open_kinematics_amd.batch import DeviceProgram

    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad", dp.kernel_note
    return dp


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("mode", ["pinned", "softnorm"])
def test_device_tangents_match_the_reference(golden, name, mode):
    """Velocities at the reference's own solved states: <= 1e-9 (mm per mm of target)."""
    _, program = golden(name)
    dp = _dp(program.with_line_mode(mode))
    tg = _tg(name)
    if mode == "softnorm" and any(int(t) == 8 for t in program.row_type):
        pytest.skip("the zero-gradient line row leaves J^T J singular without the pins (DESIGN.md §4)")
    tan, tinfo = dp.tangents(tg["pos"])
    torch.cuda.synchronize()
    info = dp.tangent_info(tinfo)
    assert np.all(info["flags"] == 1)
    assert np.max(np.abs(tan.cpu().numpy() - tg["vel"])) <= 1e-9
    # the pivots of the LDL^T of the SPD matrix J^T J lie inside its spectrum [s_min^2, s_max^2]
    if mode == "softnorm":
        assert np.all(info["min_pivot"] >= tg["smallest_sv"] ** 2 * (1 - 1e-9))
        assert np.all(info["max_pivot"] <= (tg["smallest_sv"] * tg["cond"]) ** 2 * (1 + 1e-9))
    assert np.all(info["min_pivot"] > 0)


def test_device_tangents_of_device_states_match_oracle_and_finite_differences(golden):
    from oracle.oracle import Oracle

    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    t = arrays["targets_abs"]
    res = dp.solve(t)
    tan, tinfo = dp.tangents(res.positions)
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    tan = tan.cpu().numpy()
    pos = res.positions.cpu().numpy()
    free_out = [list(pinned.out_point).index(int(p)) for p in pinned.free_point]
    orc = Oracle(pinned)
    for k in (0, 37, 100):
        vel, _, _ = orc.tangents(pos[k][free_out].reshape(-1))
        assert np.max(np.abs(tan[k] - vel[:, pinned.out_point])) <= 1e-9
    # central differences of device solves, as tests/test_sensitivity.py:39-80 does with FD_STEP = 0.25
    h = 0.25
    for j in range(pinned.n_targets):
        dt = np.zeros_like(t)
        dt[:, j] = h
        fd = (dp.solve(t + dt).positions - dp.solve(t - dt).positions).cpu().numpy() / (2 * h)
        assert np.allclose(tan[:, j], fd, rtol=1e-3, atol=1e-5)


def test_ensemble_tangents_use_per_geometry_tables(golden):
    arrays, program = golden("c5_ensemble")
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    gpos, gparam = dp.rebind(torch.as_tensor(arrays["hardpoints"], device="cuda:0"))
    g, s = arrays["targets_abs"].shape[:2]
    t = torch.as_tensor(arrays["targets_abs"].reshape(g * s, -1), device="cuda:0")
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s)
    res = dp.solve(t, **kw)
    tan, tinfo = dp.tangents(res.positions, **kw)
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    h = 0.25
    dt = torch.zeros_like(t)
    dt[:, 1] = h
    fd = (dp.solve(t + dt, **kw).positions - dp.solve(t - dt, **kw).positions) / (2 * h)
    assert torch.allclose(tan[:, 1], fd, rtol=1e-3, atol=1e-5)
    # a wrong geometry table must change the answer (the tables are really read)
    wrong = dp.tangents(res.positions, geom_pos=gpos.flip(0).contiguous(), geom_row_param=gparam.flip(0).contiguous(),
                        steps_per_geometry=s)[0]
    assert float((wrong - tan).abs().max()) > 1e-6


def test_axle_tangents_in_pair_mode_and_from_the_generic_kernel_match_the_reference(golden, monkeypatch):
    """Rocker axle (n = 60, T = 3): the generated pair-mode tangent kernel (one quad per corner, the rack row through
    Sherman-Morrison) and the interpreter's one-wavefront-per-state kernel both reproduce the reference's velocities."""
    from open_kinematics_amd.batch import DeviceProgram

    _, program = golden("c3_axle_grid")
    tg = _tg("c3_axle_grid")
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    assert dp.kernel == "quad"  # pair-mode kernels
    tan, tinfo = dp.tangents(tg["pos"])
    torch.cuda.synchronize()
    info = dp.tangent_info(tinfo)
    assert np.all(info["flags"] == 1) and np.all(info["min_pivot"] > 0)
    assert np.max(np.abs(tan.cpu().numpy() - tg["vel"])) <= 1e-9
    monkeypatch.setenv("OKX_DEV", "tangent_generic")
    wave, winfo = dp.tangents(tg["pos"])
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(winfo)["flags"] == 1)
    assert np.max(np.abs(wave.cpu().numpy() - tg["vel"])) <= 1e-9
    assert float((tan - wave).abs().max()) <= 1e-10
    monkeypatch.delenv("OKX_DEV")
    # per-geometry tables; the batch is ragged anyway (fewer states than the 8 of a wavefront)
    n = tg["pos"].shape[0]
    assert n % 8 != 0
    pos = torch.as_tensor(tg["pos"], device="cuda:0")
    gpos, gparam = dp.rebind(torch.as_tensor(np.repeat(program.design_pos[None], n, axis=0)))
    per_geom, _ = dp.tangents(pos, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=1)
    assert float((per_geom - tan).abs().max()) <= 1e-12


def test_generic_and_generated_tangent_kernels_agree(golden, monkeypatch):
    arrays, program = golden("c4_macpherson_grid")
    dp = _dp(program.with_line_mode("pinned"))
    tg = _tg("c4_macpherson_grid")
    quad, _ = dp.tangents(tg["pos"])
    monkeypatch.setenv("OKX_DEV", "tangent_generic")
    wave, tinfo = dp.tangents(tg["pos"])
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    assert float((quad - wave).abs().max()) <= 1e-11
    assert np.max(np.abs(wave.cpu().numpy() - tg["vel"])) <= 1e-9


def test_sensitivity_dropin_mirrors_the_reference_module(golden):
    """compute_sweep_tangents / compute_state_tangents with the reference's call shapes and result types
    (core/sweep.py:113-141, core/sensitivity.py:57-143; tests/test_sensitivity.py:39-87)."""
    import yaml

    from open_kinematics_amd.enums import PointID
    from open_kinematics_amd.input import build_sweep, load_geometry
    from open_kinematics_amd.sensitivity import (TangentField, TangentSolveInfo, combine_tangents,
                                                 compute_state_tangents, compute_sweep_tangents)
    from open_kinematics_amd.solver import convert_targets_to_absolute
    from open_kinematics_amd.sweep import solve_sweep

    arrays, program = golden("c1_dw_corner")
    tg = _tg("c1_dw_corner")
    sus = load_geometry(os.path.join(GOLDEN, "geometry", "geometry.yaml"))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    states, _ = solve_sweep(sus, sweep)
    tangents = compute_sweep_tangents(sus, sweep, states)
    assert len(tangents.per_step) == len(tangents.solve_infos) == len(states)
    out = sus.output_points()
    for j, step in enumerate(tg["step_index"]):
        fields = tangents.per_step[step]
        assert len(fields) == 2 and isinstance(fields[0], TangentField) and fields[1].target_index == 1
        got = np.array([[f.velocity(k) for k in out] for f in fields])
        assert np.max(np.abs(got - tg["vel"][j])) <= 1e-7  # device states vs the reference's states
        info = tangents.solve_infos[step]
        assert isinstance(info, TangentSolveInfo) and not info.rank_deficient and info.rank == info.n_variables == 18
        assert info.smallest_singular_value > 0.0 and np.isfinite(info.condition_number)
    assert tangents.per_step[50][1].velocity(PointID.WHEEL_CENTER)[2] == pytest.approx(1.0)
    assert np.array_equal(tangents.per_step[0][0].velocity("no such point"), np.zeros(3))
    # single-state form
    initial = sus.initial_state()
    step_targets = convert_targets_to_absolute([s[50] for s in sweep.target_sweeps], initial)
    fields, info = compute_state_tangents(states[50], sus.constraints(), sus.derived_spec(), step_targets)
    for f, g in zip(fields, tangents.per_step[50]):
        for key in states[50].positions:
            assert np.max(np.abs(f.velocity(key) - g.velocity(key))) <= 1e-12
    assert compute_state_tangents(states[0], sus.constraints(), sus.derived_spec(), [])[0] == []
    both = combine_tangents(fields, [2.0, -1.0])
    assert np.allclose(both[PointID.WHEEL_CENTER], 2.0 * fields[0].velocity(PointID.WHEEL_CENTER)
                       - fields[1].velocity(PointID.WHEEL_CENTER))
    with pytest.raises(ValueError, match="Field/coefficient count mismatch"):
        combine_tangents(fields, [1.0])
