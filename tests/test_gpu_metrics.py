"""GPU: corner state metrics + derivative columns from okx_corner_metrics_batch against the reference."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, gpu_available
from test_metrics_oracle import (
    ANTI_FIXTURES,
    close,
    derivative_plan,
    geometry_kwargs,
    load_metrics_golden,
    out_names,
    role_indices,
    role_indices_by_name,
)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _roles_from(names, design, mg, prefix="", side_tag=""):
    """okx_corner_roles from a metrics golden's role names (design = design positions of the output points)."""
    from open_kinematics_amd.metrics import roles_from_arrays

    return roles_from_arrays(names, design, mg, prefix, side_tag)


def _roles(program, mg):
    return _roles_from(out_names(program), program.design_pos[program.out_point], mg)


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "e2e_sweep"])
def test_device_metrics_and_derivatives_match_the_reference(golden, name):
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_state_metrics

    _, program = golden(name)
    pinned = program.with_line_mode("pinned")
    mg = load_metrics_golden(name)
    roles, ridx = _roles(pinned, mg)
    dp = DeviceProgram(pinned, "cuda:0")
    pos = torch.as_tensor(mg["pos"], device="cuda:0")
    tan, tinfo = dp.tangents(pos)
    res = corner_state_metrics(roles, pos, tan)
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    values = res.values.cpu().numpy()
    assert np.max(np.abs(values[:, :8] - mg["values"][:, :8])) <= 1e-9
    assert close(values, mg["values"], 1e-9)  # NaN exactly where the reference reports None
    deriv = res.derivatives.cpu().numpy()
    tan = tan.cpu().numpy()
    plan = derivative_plan(pinned, mg["deriv_names"])
    assert len(plan) >= 8
    for j, (what, t) in plan.items():
        got = tan[:, t, ridx["wheel_center"], what[1]] if isinstance(what, tuple) else deriv[:, t, what]
        ref = mg["deriv"][:, j]
        assert np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-7, mg["deriv_names"][j]
    assert res.column("camber").shape == (pos.shape[0],) and METRIC_NAMES[0] == "camber"


def test_metrics_from_the_suspension_role_hooks(golden):
    """corner_roles() from the loader's own role hooks; solve -> tangents -> metrics stays on the device."""
    import yaml

    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import build_sweep, load_geometry
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_roles, corner_state_metrics
    from open_kinematics_amd.sweep import sweep_program

    arrays, _ = golden("c4_macpherson_grid")
    mg = load_metrics_golden("c4_macpherson_grid")
    sus = load_geometry(os.path.join(GOLDEN, "geometry", "macpherson_geometry.yaml"))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    program, targets = sweep_program(sus, sweep)
    dp = DeviceProgram(program, "cuda:0")
    res = dp.solve(torch.as_tensor(targets, device="cuda:0"))
    tan, _ = dp.tangents(res.positions)
    m = corner_state_metrics(corner_roles(sus, program), res.positions, tan)
    torch.cuda.synchronize()
    values = m.values.cpu().numpy()
    assert np.max(np.abs(values[:, :8] - mg["values"][:, :8])) <= 5e-5  # device states vs the reference's default-tolerance states (1.4e-5 mm apart)
    assert close(values, mg["values"], 1e-4)  # the instant-centre family amplifies that state difference
    assert m.derivatives.shape == (targets.shape[0], 2, len(METRIC_NAMES))
    with pytest.raises(ValueError, match="no tangents"):
        corner_state_metrics(corner_roles(sus, program), res.positions).derivative("camber", 0)


@pytest.mark.parametrize("name", sorted(ANTI_FIXTURES))
def test_anti_geometry_matches_the_reference(golden, name):
    """Brake bias / axle position / driven axle authored: anti-dive, anti-lift and anti-squat are defined."""
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_state_metrics

    _, program = golden(ANTI_FIXTURES[name])
    mg = load_metrics_golden(name)
    roles, _ = _roles(program, mg)
    res = corner_state_metrics(roles, torch.as_tensor(mg["pos"], device="cuda:0"))
    torch.cuda.synchronize()
    values = res.values.cpu().numpy()
    assert close(values, mg["values"], 1e-9)
    assert np.isfinite(values[:, METRIC_NAMES.index("anti_squat")]).all()
    assert np.isfinite(values[:, METRIC_NAMES.index("anti_dive" if name == "dw_front_anti" else "anti_lift")]).all()


def test_axle_metrics_and_both_corner_rows_match_the_reference(golden):
    from open_kinematics_amd.metrics import AXLE_METRIC_NAMES, axle_state_metrics, corner_state_metrics

    _, program = golden("c3_axle_grid")
    mg = load_metrics_golden("axle_c3")
    names = out_names(program)
    design = program.design_pos[program.out_point]
    pos = torch.as_tensor(mg["pos"], device="cuda:0")
    left, _ = _roles_from(names, design, mg, "left_", "left_")
    right, _ = _roles_from(names, design, mg, "right_", "right_")
    axle = axle_state_metrics(left, right, pos).cpu().numpy()
    assert axle.shape == (pos.shape[0], len(AXLE_METRIC_NAMES))
    assert close(axle, mg["axle_values"], 1e-9)
    for roles, tag in ((left, "left"), (right, "right")):
        values = corner_state_metrics(roles, pos).values.cpu().numpy()
        assert close(values, mg[f"{tag}_values"], 1e-9), tag


def test_axle_roles_from_the_loader_hooks(golden):
    """axle_roles() over the loader's axle: solve the axle grid, metrics of the device states vs the reference."""
    import yaml

    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import axle_roles, axle_state_metrics, corner_state_metrics
    from open_kinematics_amd.sweep import sweep_program

    arrays, _ = golden("c3_axle_grid")
    mg = load_metrics_golden("axle_c3")
    axle = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), axle)
    program, targets = sweep_program(axle, sweep)
    dp = DeviceProgram(program, "cuda:0")
    stride = 5  # oracle/gen_golden_metrics.py: emit_axle("c3_axle_grid", "axle_c3", stride=5)
    res = dp.solve(torch.as_tensor(targets[::stride].copy(), device="cuda:0"))
    left, right = axle_roles(axle, program)
    got = axle_state_metrics(left, right, res.positions).cpu().numpy()
    ref = mg["axle_values"]
    assert got.shape == ref.shape
    # device states vs the reference's default-tolerance states; the roll centre amplifies the difference
    assert np.max(np.abs(got[:, :4] - ref[:, :4])) <= 5e-5 and close(got[:, 4:6], ref[:, 4:6], 1e-4)
    assert np.max(np.abs(got[:, 6] - ref[:, 6])) <= 5e-5
    values = corner_state_metrics(left, res.positions).values.cpu().numpy()
    assert np.max(np.abs(values[:, :8] - mg["left_values"][:, :8])) <= 5e-5


def test_axle_topology_rotations_and_their_derivative_columns(golden):
    """rocker_angle / torsion_bar_twist / arb_arm_angle / arb_twist and deriv_*_wrt_hub_z(_left|_right) on the device."""
    import yaml

    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import axle_topology_metrics, topology_rotation_roles
    from open_kinematics_amd.sweep import sweep_program

    arrays, program = golden("c3_axle_grid")
    mg = load_metrics_golden("axle_c3")
    axle = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    pinned, _ = sweep_program(axle, build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), axle))
    assert list(pinned.out_point) == list(program.out_point)  # the loader's program = the golden's (typed keys)
    names, roles = topology_rotation_roles(axle, pinned)
    assert names == ["rocker_angle_left", "torsion_bar_twist_left", "rocker_angle_right", "torsion_bar_twist_right",
                     "arb_arm_angle_left", "arb_arm_angle_right"]
    dp = DeviceProgram(pinned, "cuda:0")
    pos = torch.as_tensor(mg["pos"], device="cuda:0")
    tan, tinfo = dp.tangents(pos)
    m = axle_topology_metrics(axle, pinned, pos, tan)
    torch.cuda.synchronize()
    assert np.all(dp.tangent_info(tinfo)["flags"] == 1)
    for tag in ("left", "right"):
        for j, name in enumerate(str(n) for n in mg[f"{tag}_extra_names"]):
            assert np.max(np.abs(m[f"{name}_{tag}"].cpu().numpy() - mg[f"{tag}_extra_values"][:, j])) <= 1e-9, (tag, name)
    assert np.max(np.abs(m["arb_twist"].cpu().numpy() - mg["axle_extra_values"][:, 0])) <= 1e-9
    # derivative columns: d / d (hub z of one side), the other hub and the rack held
    from open_kinematics_amd.results_writer import point_key_name

    tnames = [point_key_name(pinned.point_keys[p]) for p in pinned.tgt_point]
    hub = {tag: next(t for t, (n, d) in enumerate(zip(tnames, pinned.tgt_dir)) if n == f"{tag}_wheel_center" and d[2] == 1.0)
           for tag in ("left", "right")}
    checked = 0
    for tag in ("left", "right"):
        for j, col in enumerate(str(c) for c in mg[f"{tag}_deriv_names"]):
            response, driver = col[len("deriv_"):].split("_wrt_")
            if driver == "hub_z" and f"d_{response}_{tag}" in m:
                got = m[f"d_{response}_{tag}"][:, hub[tag]].cpu().numpy()
                ref = mg[f"{tag}_deriv"][:, j]
                assert np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-7, (tag, col)
                checked += 1
    for j, col in enumerate(str(c) for c in mg["axle_deriv_names"]):
        tag = col.rsplit("_", 1)[1]
        got = m["d_arb_twist"][:, hub[tag]].cpu().numpy()
        assert np.max(np.abs(got - mg["axle_deriv"][:, j]) / np.maximum(1.0, np.abs(mg["axle_deriv"][:, j]))) <= 1e-7, col
        checked += 1
    assert checked == 6  # rocker_angle + torsion_bar_twist per side, arb_twist wrt each hub


def test_bad_roles_are_rejected(golden):
    from open_kinematics_amd.metrics import axle_state_metrics, corner_state_metrics, make_roles

    base = dict(wheel_center=0, contact_patch=1, axle_inboard=2, axle_outboard=3, steer_lower=4, steer_upper=5,
                side_sign=1.0, design_wheel_center_z=0.0)
    pos = torch.zeros((4, 15, 3), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="not an output point"):
        corner_state_metrics(make_roles(**{**base, "wheel_center": 99}), pos)
    with pytest.raises(ValueError, match="side_sign"):
        corner_state_metrics(make_roles(**{**base, "side_sign": 0.5}), pos)
    with pytest.raises(ValueError, match="instant-axis point"):
        corner_state_metrics(make_roles(**base, instant_axis=("two_planes", (0, 1, 2, 3, 4, 15))), pos)
    with pytest.raises(ValueError, match="damper"):
        corner_state_metrics(make_roles(**base, damper=(3, 44)), pos)
    with pytest.raises(ValueError, match="right"):
        axle_state_metrics(make_roles(**base), make_roles(**{**base, "side_sign": 0.0}), pos)
    from open_kinematics_amd.metrics import axis_rotation_metrics, rotation_role

    with pytest.raises(ValueError, match="rotation 0: not an output point"):
        axis_rotation_metrics([rotation_role(77, (1, 0, 0), (0, 0, 0), (0, 0, 1))], pos)
    with pytest.raises(ValueError, match="distinct"):
        rotation_role(0, (1, 0, 0), (0, 0, 0), (0, 0, 0))
    on_axis = axis_rotation_metrics([rotation_role(0, (0, 0, 5), (0, 0, 0), (0, 0, 1))], pos)[0]
    assert torch.isnan(on_axis).all()  # the reference raises for a point on its rotation axis
    # no instant-axis construction, no damper, no vehicle numbers: those columns read NaN, the rest are defined
    pos = torch.rand((4, 15, 3), dtype=torch.float64, device="cuda:0")
    values = corner_state_metrics(make_roles(**base), pos).values.cpu().numpy()
    assert np.isfinite(values[:, :8]).all() and np.isnan(values[:, 8:]).all()


def test_tiled_kernel_gives_the_bits_of_the_one_thread_per_state_kernel(golden):
    """
    okx_corner_metrics_batch stages records of up to 21 points through LDS, 64 states per wavefront; longer records run
    one thread per state.  Same arithmetic: the same states padded to 22-point records must give the same bits - values,
    derivative columns, NaN pattern - for ragged batch sizes, one state, and bases that are not 16-byte aligned.
    """
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles, corner_state_metrics
    from open_kinematics_amd.workloads import bump_sweep_problem, geometry_path

    program, targets = bump_sweep_problem(1000)
    roles = corner_roles(load_geometry(geometry_path("geometry.yaml")), program)
    dp = DeviceProgram(program, "cuda:0")
    pos = dp.solve(torch.as_tensor(targets, device="cuda:0"), chain_len=-1).positions
    tan, _ = dp.tangents(pos)
    n_out, T = pos.shape[1], tan.shape[1]
    assert 3 * n_out <= 63  # the tiled path
    wide = torch.zeros((pos.shape[0], 22, 3), dtype=torch.float64, device="cuda:0")
    wide[:, :n_out] = pos
    wide_tan = torch.zeros((pos.shape[0], T, 22, 3), dtype=torch.float64, device="cuda:0")
    wide_tan[:, :, :n_out] = tan
    ref = corner_state_metrics(roles, wide, wide_tan)
    ref_v, ref_d = ref.values.cpu().numpy(), ref.derivatives.cpu().numpy()
    assert np.isfinite(ref_v[:, :8]).all() and np.isfinite(ref_d[:, :, :8]).all()

    def same(a, b):
        return np.array_equal(a.view(np.int64), b.view(np.int64))

    for lo, hi in ((0, 1000), (0, 1), (0, 63), (0, 64), (0, 65), (1, 130), (7, 8), (333, 1000)):
        got = corner_state_metrics(roles, pos[lo:hi], tan[lo:hi])  # (an odd `lo`: a base that is 8-byte aligned only)
        assert same(got.values.cpu().numpy(), ref_v[lo:hi]), (lo, hi)
        assert same(got.derivatives.cpu().numpy(), ref_d[lo:hi]), (lo, hi)
        alone = corner_state_metrics(roles, pos[lo:hi], None)
        assert alone.derivatives is None and same(alone.values.cpu().numpy(), ref_v[lo:hi]), (lo, hi)
