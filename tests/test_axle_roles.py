"""
CPU: the host side of the composed axle's evaluated solve - ``metrics.axle_evaluation_roles`` (both corners' roles, the
DISTINCT rotation / hardware roles and the column every reference metric name reads) and the argument checks of
``okx_precompile_axle_evaluation`` / ``okx_program_enable_axle_evaluation`` (``include/okx.h``; no device needed).
"""

import ctypes as C

import numpy as np
import pytest

from open_kinematics_amd import _abi, _lib


def _axle(name: str = "axle_geometry_rocker.yaml"):
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.workloads import axle_grid_problem, geometry_path

    program, _ = axle_grid_problem(3, 3)
    return load_geometry(geometry_path(name)), program


def test_axle_roles_hold_both_corners_and_the_distinct_rotation_roles():
    from open_kinematics_amd.enums import PointID, PointRef, Side
    from open_kinematics_amd.metrics import AxleRoles, CornerRoles, RotationRole, axle_evaluation_roles

    axle, program = _axle()
    roles, rot_names, hw_names = axle_evaluation_roles(axle, program)
    assert isinstance(roles, AxleRoles) and C.sizeof(AxleRoles) == 2 * C.sizeof(CornerRoles) + 8 + 8 * C.sizeof(RotationRole)
    out_keys = [program.point_keys[k] for k in program.out_point]
    assert out_keys[roles.left.wheel_center] == PointRef(Side.LEFT, PointID.WHEEL_CENTER)
    assert out_keys[roles.right.wheel_center] == PointRef(Side.RIGHT, PointID.WHEEL_CENTER)
    assert roles.left.side_sign == 1.0 and roles.right.side_sign == -1.0
    # rocker angle and torsion-bar twist are ONE role per side (the same rotation), the U-bar's arm angles one each
    assert rot_names == ["rocker_angle_left", "torsion_bar_twist_left", "rocker_angle_right", "torsion_bar_twist_right",
                         "arb_arm_angle_left", "arb_arm_angle_right"] and hw_names == []
    assert roles.n_roles == 4
    assert roles.column_of["rocker_angle_left"] == roles.column_of["torsion_bar_twist_left"] == 0
    assert roles.column_of["rocker_angle_right"] == 1 and roles.column_of["arb_arm_angle_left"] == 2 and roles.column_of["arb_arm_angle_right"] == 3
    # partners of one kind sit side by side (the kernel evaluates such a pair on its two quads at once)
    assert [roles.roles[k].kind for k in range(4)] == [0, 0, 0, 0]
    assert out_keys[roles.roles[0].point] == PointRef(Side.LEFT, PointID.PUSHROD_INBOARD)
    assert out_keys[roles.roles[3].point] == PointRef(Side.RIGHT, PointID.DROPLINK_U_BAR)


def test_more_than_eight_distinct_roles_are_refused(monkeypatch):
    from open_kinematics_amd import metrics

    axle, program = _axle()
    names, roles = metrics.topology_rotation_roles(axle, program)

    def many(_axle, _program):
        extra = []
        for k in range(6):
            role = metrics.rotation_role(roles[0].point, (1.0 + k, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
            extra.append(role)
        return [f"extra_{k}" for k in range(6)], extra

    monkeypatch.setattr(metrics, "hardware_roles", many)
    with pytest.raises(ValueError, match="at most 8"):
        metrics.axle_evaluation_roles(axle, program)


def test_precompile_checks_its_arguments_without_a_device():
    from open_kinematics_amd.metrics import axle_evaluation_roles
    from open_kinematics_amd.workloads import bump_sweep_problem

    lib = _lib.load()
    axle, program = _axle()
    roles, _, _ = axle_evaluation_roles(axle, program)
    host = _abi.HostProgram(program)
    bad = type(roles).from_buffer_copy(bytes(roles))
    bad.n_roles = 9
    assert lib.okx_precompile_axle_evaluation(host.byref(), C.byref(bad)) == -1 and b"n_roles" in lib.okx_last_error()
    bad = type(roles).from_buffer_copy(bytes(roles))
    bad.roles[1].point = program.n_out
    assert lib.okx_precompile_axle_evaluation(host.byref(), C.byref(bad)) == -1 and b"outside the output list" in lib.okx_last_error()
    bad = type(roles).from_buffer_copy(bytes(roles))
    bad.right.instant_axis_kind = 0                       # the two corners must share the instant-axis construction
    assert lib.okx_precompile_axle_evaluation(host.byref(), C.byref(bad)) == -1 and b"instant-axis" in lib.okx_last_error()
    bad = type(roles).from_buffer_copy(bytes(roles))
    bad.left.wheel_center = -1
    assert lib.okx_precompile_axle_evaluation(host.byref(), C.byref(bad)) == -1 and b"left" in lib.okx_last_error()
    # a single-mode program (one corner) has no pair-mode module: OKX_ERR_LIMIT, the corner entry point is the one to use
    corner, _ = bump_sweep_problem(4)
    small = type(roles).from_buffer_copy(bytes(roles))
    for side in (small.left, small.right):
        for field in ("wheel_center", "contact_patch", "axle_inboard", "axle_outboard", "steer_lower", "steer_upper"):
            setattr(side, field, 0)
        side.instant_axis_kind, side.damper_top, side.damper_bottom, side.rack_attachment = 0, -1, -1, -1
    small.n_roles = 0
    assert lib.okx_precompile_axle_evaluation(_abi.HostProgram(corner).byref(), C.byref(small)) == -2
    assert b"pair-mode program" in lib.okx_last_error()
    assert lib.okx_program_eval_columns(None) == 0
