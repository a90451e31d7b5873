"""GPU: corner state metrics + derivative columns from okx_corner_metrics_batch against the reference."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, gpu_available
from test_metrics_oracle import derivative_plan, load_metrics_golden, role_indices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _roles(program, mg):
    from open_kinematics_amd.metrics import CornerRoles

    r = role_indices(program, mg)
    design_z = float(program.design_pos[program.out_point[r["wheel_center"]]][2])
    return CornerRoles(r["wheel_center"], r["contact_patch"], r["axle_inboard"], r["axle_outboard"],
                       r["steer_lower"], r["steer_upper"], float(mg["side_sign"]), design_z), r


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
    assert np.max(np.abs(values - mg["values"])) <= 1e-9
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
    from open_kinematics_amd.metrics import corner_roles, corner_state_metrics
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
    assert np.max(np.abs(m.values.cpu().numpy() - mg["values"])) <= 5e-5  # device states vs the reference's default-tolerance states (1.4e-5 mm apart)
    assert m.derivatives.shape == (targets.shape[0], 2, 8)
    with pytest.raises(ValueError, match="no tangents"):
        corner_state_metrics(corner_roles(sus, program), res.positions).derivative("camber", 0)


def test_bad_roles_are_rejected(golden):
    from open_kinematics_amd.metrics import CornerRoles, corner_state_metrics

    pos = torch.zeros((4, 15, 3), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="not an output point"):
        corner_state_metrics(CornerRoles(99, 0, 1, 2, 3, 4, 1.0, 0.0), pos)
    with pytest.raises(ValueError, match="side_sign"):
        corner_state_metrics(CornerRoles(0, 1, 2, 3, 4, 5, 0.5, 0.0), pos)
