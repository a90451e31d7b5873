"""
State metrics of solved sweep states on the device: the reference's whole corner
catalog (``core/metrics/catalog.py``: angles, travel, steering geometry, instant centres, swing arms,
damper length, anti-geometry) plus the axle-scope metrics (``axle_metrics.py``), and their derivatives along
the solution-manifold tangents (the raw material of ``core/metrics/derivatives.py``'s
``deriv_<response>_wrt_<driver>`` columns).  One streaming kernel launch per batch
(``okx_corner_metrics_batch``); tensors stay in HBM.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .enums import PointID, PointRef, Side

METRIC_NAMES = (
    "camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius", "mechanical_trail",
    "svic_x", "svic_z", "svsa_length", "fvic_y", "fvic_z", "fvsa_length", "damper_length", "svsa_angle",
    "anti_dive", "anti_lift", "anti_squat",
)  # column order = OKX_METRIC_* (include/okx.h)
# the reference's export order (metrics/catalog.py:71-146)
CATALOG_ORDER = (
    "camber", "caster", "kpi", "scrub_radius", "mechanical_trail", "roadwheel_angle", "svic_x", "svic_z",
    "svsa_length", "fvic_y", "fvic_z", "fvsa_length", "wheel_travel", "half_track", "damper_length", "svsa_angle",
    "anti_dive", "anti_lift", "anti_squat",
)
AXLE_METRIC_NAMES = ("heave", "roll", "ride_height_change", "track", "roll_center_y", "roll_center_z",
                     "rack_displacement")  # = OKX_AXLE_METRIC_*, the order of metrics/axle_metrics.py:47-70
_IA_KINDS = {None: 0, "two_planes": 1, "plane_and_strut": 2}
_AXLE_POSITIONS = {None: 0, "front": 1, "rear": 2}


class CornerRoles(C.Structure):
    """ctypes mirror of ``okx_corner_roles``."""

    _fields_ = [
        ("wheel_center", C.c_int32), ("contact_patch", C.c_int32),
        ("axle_inboard", C.c_int32), ("axle_outboard", C.c_int32),
        ("steer_lower", C.c_int32), ("steer_upper", C.c_int32),
        ("instant_axis_kind", C.c_int32), ("instant_axis_point", C.c_int32 * 6),
        ("damper_top", C.c_int32), ("damper_bottom", C.c_int32),
        ("rack_attachment", C.c_int32),
        ("axle_position", C.c_int32), ("driven_axle", C.c_int32),
        ("side_sign", C.c_double), ("design_wheel_center_z", C.c_double),
        ("design_contact_patch_z", C.c_double), ("design_rack_y", C.c_double),
        ("wheelbase", C.c_double), ("cg_z", C.c_double), ("front_brake_bias", C.c_double),
    ]


def make_roles(*, wheel_center: int, contact_patch: int, axle_inboard: int, axle_outboard: int, steer_lower: int,
               steer_upper: int, side_sign: float, design_wheel_center_z: float,
               instant_axis: tuple | None = None, damper: tuple | None = None, rack_attachment: int = -1,
               design_contact_patch_z: float = 0.0, design_rack_y: float = 0.0, wheelbase: float = float("nan"),
               cg_z: float = float("nan"), front_brake_bias: float | None = None, axle_position: str | None = None,
               driven_axle: str | None = None) -> CornerRoles:
    """
    ``okx_corner_roles`` from plain indices.  ``instant_axis`` = ``("two_planes", six indices)`` or
    ``("plane_and_strut", four indices)``; ``damper`` = ``(top, bottom)``.
    """
    kind, points = instant_axis if instant_axis is not None else (None, ())
    if kind not in _IA_KINDS:
        raise ValueError(f"unknown instant-axis construction {kind!r}")
    if axle_position not in _AXLE_POSITIONS or driven_axle not in _AXLE_POSITIONS:
        raise ValueError("axle_position / driven_axle must be None, 'front' or 'rear'")
    padded = [int(k) for k in points] + [-1] * (6 - len(points))
    top, bottom = damper if damper is not None else (-1, -1)
    return CornerRoles(
        wheel_center=wheel_center, contact_patch=contact_patch, axle_inboard=axle_inboard, axle_outboard=axle_outboard,
        steer_lower=steer_lower, steer_upper=steer_upper, instant_axis_kind=_IA_KINDS[kind],
        instant_axis_point=(C.c_int32 * 6)(*padded), damper_top=int(top), damper_bottom=int(bottom),
        rack_attachment=int(rack_attachment), axle_position=_AXLE_POSITIONS[axle_position],
        driven_axle=_AXLE_POSITIONS[driven_axle], side_sign=float(side_sign),
        design_wheel_center_z=float(design_wheel_center_z), design_contact_patch_z=float(design_contact_patch_z),
        design_rack_y=float(design_rack_y), wheelbase=float(wheelbase), cg_z=float(cg_z),
        front_brake_bias=float("nan") if front_brake_bias is None else float(front_brake_bias),
    )


def roles_from_arrays(names, design, arrays, prefix: str = "", side_tag: str = "") -> tuple[CornerRoles, dict]:
    """
    ``okx_corner_roles`` from role NAMES kept as arrays (``roles`` = wheel-axis inboard / outboard and steering-axis lower /
    upper point names, ``axis_kind`` / ``axis_points``, ``damper``, ``rack``, ``side_sign`` and the vehicle numbers - the
    layout of the committed metric fixtures, ``oracle/gen_golden_metrics.py``): ``names`` = lower-case names of the
    program's output points, ``design`` their design positions.  Returns the roles and the six basic role indices.
    Used by the parity tests and by ``__graft_entry__.build()`` to precompile exactly the modules those tests ask for.
    """
    axle_in, axle_out, lower, upper = (side_tag + str(v).lower() for v in arrays[prefix + "roles"])
    r = {"wheel_center": names.index(side_tag + "wheel_center"), "contact_patch": names.index(side_tag + "contact_patch_center"),
         "axle_inboard": names.index(axle_in), "axle_outboard": names.index(axle_out),
         "steer_lower": names.index(lower), "steer_upper": names.index(upper)}
    local = [n[len(side_tag):] if side_tag and n.startswith(side_tag) else ("-" if side_tag else n) for n in names]
    text = lambda key: str(arrays[prefix + key])  # noqa: E731
    damper = [local.index(str(n)) for n in arrays[prefix + "damper"]]
    bias = float(arrays[prefix + "front_brake_bias"])
    rack = text("rack") if prefix + "rack" in arrays else ""
    rack_idx = names.index(side_tag + rack) if rack else -1
    roles = make_roles(
        **r, side_sign=float(arrays[prefix + "side_sign"]), design_wheel_center_z=float(design[r["wheel_center"]][2]),
        instant_axis=(text("axis_kind"), [local.index(str(n)) for n in arrays[prefix + "axis_points"]]),
        damper=damper or None, rack_attachment=rack_idx, design_contact_patch_z=float(design[r["contact_patch"]][2]),
        design_rack_y=float(design[rack_idx][1]) if rack else 0.0, wheelbase=float(arrays[prefix + "wheelbase"]),
        cg_z=float(arrays[prefix + "cg_z"]), front_brake_bias=None if np.isnan(bias) else bias,
        axle_position=text("axle_position") or None, driven_axle=text("driven_axle") or None)
    return roles, r


def corner_roles(suspension, program, side=None) -> CornerRoles:
    """
    Role indices into ``program.out_point`` from the corner's role hooks
    (``wheel_axis_points()``, ``steering_axis_points()``, ``damper_points()``, ``rack_attachment_point()``,
    the instant-axis construction, ``side``; reference ``suspensions/corner/base.py``,
    ``metrics/context.py:82-165``).  For one corner of an axle program pass the axle's corner and its
    ``side``: the points are then looked up as ``PointRef(side, point)``.
    """
    out_keys = [program.point_keys[k] for k in program.out_point]

    def index(key) -> int:
        if side is not None:
            key = PointRef(side, key)
        try:
            return out_keys.index(key)
        except ValueError:
            raise ValueError(f"metric role point {key!r} is not among the program's output points") from None

    axle_in, axle_out = suspension.wheel_axis_points()
    lower, upper = suspension.steering_axis_points()
    sign = getattr(suspension, "lateral_sign", None)
    if sign is None:
        sign = suspension.side.lateral_sign
    design = suspension.initial_state().positions
    xyz = lambda key: np.asarray(getattr(design[key], "data", design[key]), dtype=np.float64)  # noqa: E731
    axis = suspension.instant_axis_points() if hasattr(suspension, "instant_axis_points") else None
    damper = suspension.damper_points() if hasattr(suspension, "damper_points") else None
    rack = suspension.rack_attachment_point() if hasattr(suspension, "rack_attachment_point") else None
    vehicle = getattr(suspension, "vehicle", None)
    return make_roles(
        wheel_center=index(PointID.WHEEL_CENTER), contact_patch=index(PointID.CONTACT_PATCH_CENTER),
        axle_inboard=index(axle_in), axle_outboard=index(axle_out), steer_lower=index(lower), steer_upper=index(upper),
        side_sign=float(sign), design_wheel_center_z=float(xyz(PointID.WHEEL_CENTER)[2]),
        instant_axis=None if axis is None else (axis[0], tuple(index(k) for k in axis[1])),
        damper=None if damper is None else (index(damper[0]), index(damper[1])),
        rack_attachment=-1 if rack is None else index(rack),
        design_contact_patch_z=float(xyz(PointID.CONTACT_PATCH_CENTER)[2]),
        design_rack_y=0.0 if rack is None else float(xyz(rack)[1]),
        wheelbase=float("nan") if vehicle is None else vehicle.wheelbase,
        cg_z=float("nan") if vehicle is None else vehicle.cg_position[2],
        front_brake_bias=None if vehicle is None else vehicle.front_brake_bias,
        axle_position=None if vehicle is None else vehicle.axle_position,
        driven_axle=None if vehicle is None else vehicle.driven_axle,
    )


class RotationRole(C.Structure):
    """ctypes mirror of ``okx_rotation_role``."""

    _fields_ = [("point", C.c_int32), ("point_b", C.c_int32), ("design", C.c_double * 3), ("axis_point", C.c_double * 3),
                ("axis_dir", C.c_double * 3), ("scale", C.c_double), ("kind", C.c_int32), ("reserved", C.c_int32)]


# okx.h OKX_ROLE_*
ROLE_AXIS_ROTATION, ROLE_MIDPOINT_ROTATION, ROLE_STEM_TWIST, ROLE_DISTANCE, ROLE_MIDPOINT_COORDINATE = range(5)


def _vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def pair_role(kind: int, point_a: int, point_b: int, *, design=(0.0, 0.0, 0.0), axis_point=(0.0, 0.0, 0.0),
              axis_dir=(1.0, 0.0, 0.0), scale: float = 1.0) -> RotationRole:
    """A role over TWO output points (``okx.h`` OKX_ROLE_* kinds 1-4: midpoint rotation, stem twist, distance, midpoint
    coordinate); ``axis_dir`` must be a unit vector."""
    return RotationRole(point=int(point_a), point_b=int(point_b), design=_vec3(design), axis_point=_vec3(axis_point),
                        axis_dir=_vec3(axis_dir), scale=float(scale), kind=int(kind))


def rotation_role(point: int, design, axis_a, axis_b, scale: float = 1.0) -> RotationRole:
    """Rotation of output point ``point`` from ``design`` about the fixed axis through ``axis_a`` towards ``axis_b``."""
    a, b = np.asarray(axis_a, dtype=np.float64), np.asarray(axis_b, dtype=np.float64)
    length = float(np.linalg.norm(b - a))
    if length < 1e-6:
        raise ValueError("rotation axis points must be distinct")
    return RotationRole(point=int(point), design=_vec3(design), axis_point=_vec3(a), axis_dir=_vec3((b - a) / length),
                        scale=float(scale), kind=ROLE_AXIS_ROTATION)


def topology_rotation_roles(suspension, program, side=None) -> tuple[list[str], list[RotationRole]]:
    """
    The fixed-axis rotation metrics a topology declares, as ``(column names, roles)``:
    a rocker-actuated corner -> ``rocker_angle`` (and ``torsion_bar_twist`` with a torsion-bar spring,
    ``corner/mechanisms.py:378-407,611-623``); an axle -> both corners' columns with ``_left`` / ``_right`` suffixes
    plus ``arb_arm_angle_left`` / ``_right`` for a U-bar (``axle/mechanisms.py:402-430``; ``arb_twist`` is their
    difference, see ``axle_topology_metrics``).
    """
    out_keys = [program.point_keys[k] for k in program.out_point]
    if hasattr(suspension, "corners"):  # axle
        names, roles = [], []
        for s in (Side.LEFT, Side.RIGHT):
            n, r = topology_rotation_roles(suspension.corners[s], program, s)
            names += [f"{k}_{s.name.lower()}" for k in n]
            roles += r
        if suspension.arb_kind == "u_bar":
            a, b = (suspension.arb_center_points[k].data for k in (PointID.ARB_U_BAR_AXIS_A, PointID.ARB_U_BAR_AXIS_B))
            for s in (Side.LEFT, Side.RIGHT):
                names.append(f"arb_arm_angle_{s.name.lower()}")
                roles.append(rotation_role(out_keys.index(PointRef(s, PointID.DROPLINK_U_BAR)),
                                           suspension.arb_droplink_points[s].data, a, b))
        return names, roles
    actuation = getattr(suspension, "actuation", None)
    if actuation is None or not actuation.rocker:
        return [], []
    design = suspension.initial_state().positions
    key = lambda p: PointRef(side, p) if side is not None else p  # noqa: E731
    role = rotation_role(out_keys.index(key(PointID.PUSHROD_INBOARD)), design[PointID.PUSHROD_INBOARD].data,
                         design[PointID.ROCKER_AXIS_A].data, design[PointID.ROCKER_AXIS_B].data, suspension.lateral_sign)
    names = ["rocker_angle"] + (["torsion_bar_twist"] if suspension.spring.kind == "torsion_bar" else [])
    return names, [role] * len(names)


def axis_rotation_metrics(roles, positions: torch.Tensor, tangents: torch.Tensor | None = None):
    """
    ``okx_axis_rotation_batch``: ``positions [B, n_out, 3]`` (+ ``tangents [B, T, n_out, 3]``) ->
    ``(angles [B, K] deg, d angles / d target [B, T, K] or None)`` for the K ``roles``.
    """
    if not positions.is_cuda:
        raise RuntimeError("axis_rotation_metrics needs device tensors (there is no CPU fallback)")
    roles = list(roles)
    lib = _lib.load()
    pos = positions.to(torch.float64).contiguous()
    b, n_out, k = pos.shape[0], pos.shape[1], len(roles)
    angles = torch.empty((b, k), dtype=torch.float64, device=pos.device)
    tan = deriv = None
    n_targets = 0
    if tangents is not None:
        tan = tangents.to(torch.float64).contiguous()
        if tan.shape[0] != b or tan.shape[2:] != (n_out, 3):
            raise ValueError("tangents must be [B, T, n_out, 3]")
        n_targets = tan.shape[1]
        deriv = torch.empty((b, n_targets, k), dtype=torch.float64, device=pos.device)
    array = (RotationRole * max(k, 1))(*roles)
    stream = torch.cuda.current_stream(pos.device).cuda_stream
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    with torch.cuda.device(pos.device):
        rc = lib.okx_axis_rotation_batch(array, k, b, n_out, n_targets, ptr(pos), ptr(tan), ptr(angles), ptr(deriv),
                                         C.c_void_p(stream))
    _lib.check(rc, "okx_axis_rotation_batch")
    return angles, deriv


def axle_topology_metrics(axle, program, positions: torch.Tensor, tangents: torch.Tensor | None = None) -> dict:
    """
    Column name -> device tensor ``[B]`` of an axle's topology-specific state metrics (``rocker_angle_left`` ...,
    ``arb_arm_angle_left`` / ``_right``, ``arb_twist`` = left - right) and, with tangents, ``d_<name>`` ->
    ``[B, T]`` derivatives with respect to every sweep target.
    """
    names, roles = topology_rotation_roles(axle, program)
    if not names:
        return {}
    angles, deriv = axis_rotation_metrics(roles, positions, tangents)
    out = {n: angles[:, k] for k, n in enumerate(names)}
    if deriv is not None:
        out.update({f"d_{n}": deriv[:, :, k] for k, n in enumerate(names)})
    if "arb_arm_angle_left" in out:
        out["arb_twist"] = out["arb_arm_angle_left"] - out["arb_arm_angle_right"]
        if deriv is not None:
            out["d_arb_twist"] = out["d_arb_arm_angle_left"] - out["d_arb_arm_angle_right"]
    return out


def hardware_roles(axle, program) -> tuple[list[str], list[RotationRole]]:
    """
    ``(column names, roles)`` of the state metrics of an axle's shared hardware that are not rotations of one point about
    a fixed axis: a rigid T-bar's ``t_bar_heave_angle`` (crossbar midpoint about the pivot's lateral axis, design ->
    current), ``arb_twist`` (crossbar rotation about the moving stem, minus its design value) and ``t_bar_center_x``, a
    rocker-to-rocker heave link's ``heave_link_length`` (``axle/mechanisms.py:718-815,903-944``) - evaluated by the
    rotation kernel (``okx_axis_rotation_batch``, role kinds 1-4) together with their rates along the tangents.
    """
    out_keys = [program.point_keys[k] for k in program.out_point]
    design = axle.initial_state().positions
    names, roles = [], []
    if getattr(axle, "arb_kind", "") == "t_bar":
        il = out_keys.index(PointRef(Side.LEFT, PointID.DROPLINK_T_BAR))
        ir = out_keys.index(PointRef(Side.RIGHT, PointID.DROPLINK_T_BAR))
        pivot = np.asarray(axle.arb_center_points[PointID.ARB_T_BAR_PIVOT].data, dtype=np.float64)
        lateral = np.array([0.0, 1.0, 0.0])
        left = np.asarray(design[PointRef(Side.LEFT, PointID.DROPLINK_T_BAR)].data, dtype=np.float64)
        right = np.asarray(design[PointRef(Side.RIGHT, PointID.DROPLINK_T_BAR)].data, dtype=np.float64)
        center = left + (right - left) / 2.0
        stem = (center - pivot) / np.linalg.norm(center - pivot)
        crossbar = (left - right) - stem * float(np.dot(left - right, stem))
        design_twist = float(np.degrees(np.arctan2(np.dot(stem, np.cross(lateral, crossbar)), crossbar[1])))  # mechanisms.py:800-815
        names += ["t_bar_heave_angle", "arb_twist", "t_bar_center_x"]
        roles += [pair_role(ROLE_MIDPOINT_ROTATION, il, ir, design=center, axis_point=pivot, axis_dir=lateral),
                  pair_role(ROLE_STEM_TWIST, il, ir, design=(design_twist, 0.0, 0.0), axis_point=pivot, axis_dir=lateral),
                  pair_role(ROLE_MIDPOINT_COORDINATE, il, ir, axis_dir=(1.0, 0.0, 0.0))]
    if getattr(axle, "heave_link", False):
        names.append("heave_link_length")
        roles.append(pair_role(ROLE_DISTANCE, out_keys.index(PointRef(Side.LEFT, PointID.HEAVE_LINK_ROCKER)),
                               out_keys.index(PointRef(Side.RIGHT, PointID.HEAVE_LINK_ROCKER))))
    return names, roles


def axle_hardware_metrics(axle, program, positions: torch.Tensor, tangents: torch.Tensor | None = None) -> dict:
    """
    The hardware metrics of ``hardware_roles`` on the device tensors the solve returned, one kernel launch: name ->
    ``[B]`` (deg / mm like the reference's rows) and, with tangents, ``d_<name>`` -> ``[B, T]`` rates along every
    target's tangent (``mechanisms.py:718-761,903-928``).
    """
    if not positions.is_cuda:
        raise RuntimeError("axle_hardware_metrics needs device tensors (there is no CPU fallback)")
    names, roles = hardware_roles(axle, program)
    if not names:
        return {}
    values, rates = axis_rotation_metrics(roles, positions, tangents)
    out = {n: values[:, k] for k, n in enumerate(names)}
    if rates is not None:
        out.update({f"d_{n}": rates[:, :, k] for k, n in enumerate(names)})
    return out


MAX_ROTATIONS = 8  # OKX_MAX_ROTATIONS


class AxleRoles(C.Structure):
    """ctypes mirror of ``okx_axle_roles``: both corners' roles and up to eight rotation / hardware roles."""

    _fields_ = [("left", CornerRoles), ("right", CornerRoles), ("n_roles", C.c_int32), ("reserved", C.c_int32),
                ("roles", RotationRole * MAX_ROTATIONS)]


def _role_key(role: RotationRole) -> tuple:
    return (role.kind, role.point, role.point_b, tuple(role.design), tuple(role.axis_point), tuple(role.axis_dir), role.scale)


def axle_evaluation_roles(axle, program) -> tuple["AxleRoles", list, list]:
    """
    What ``DeviceProgram.enable_evaluation`` takes for a composed axle: ``(okx_axle_roles, rotation names, hardware
    names)`` - both corners' roles, then the DISTINCT roles behind ``topology_rotation_roles`` (rocker angles, torsion-bar
    twists - the same role as the rocker's -, U-bar arm angles) and ``hardware_roles`` (a T-bar's heave angle, twist and
    centre travel, a heave link's length).  ``AxleRoles.column_of[name]`` is the role column of every name.
    Raises ``ValueError`` when there are more than eight distinct roles.
    """
    left, right = axle_roles(axle, program)
    rot_names, rot_roles = topology_rotation_roles(axle, program)
    hw_names, hw_roles = hardware_roles(axle, program)
    unique, column_of = [], {}
    for name, role in zip(rot_names + hw_names, rot_roles + hw_roles):
        key = _role_key(role)
        for k, other in enumerate(unique):
            if _role_key(other) == key:
                column_of[name] = k
                break
        else:
            column_of[name] = len(unique)
            unique.append(role)
    if len(unique) > MAX_ROTATIONS:
        raise ValueError(f"an evaluated axle takes at most {MAX_ROTATIONS} rotation / hardware roles, got {len(unique)}")
    # two consecutive roles of one kind are evaluated side by side (one per half): keep left / right partners adjacent
    roles = AxleRoles(left=left, right=right, n_roles=len(unique))
    for k, role in enumerate(unique):
        roles.roles[k] = role
    roles.column_of = column_of
    return roles, rot_names, hw_names


def axle_roles(axle, program) -> tuple[CornerRoles, CornerRoles]:
    """(left, right) roles of an ``AxleSuspension``'s corners inside the axle program's output points."""
    return corner_roles(axle.corners[Side.LEFT], program, Side.LEFT), corner_roles(axle.corners[Side.RIGHT], program, Side.RIGHT)


@dataclass
class CornerMetrics:
    values: torch.Tensor              # [B, 19] float64, device (NaN where the reference reports None)
    derivatives: torch.Tensor | None  # [B, T, 19]: d metric / d target (metric units per mm of target)

    def column(self, name: str) -> torch.Tensor:
        return self.values[:, METRIC_NAMES.index(name)]

    def derivative(self, name: str, target_index: int) -> torch.Tensor:
        if self.derivatives is None:
            raise ValueError("no tangents were given")
        return self.derivatives[:, target_index, METRIC_NAMES.index(name)]


def corner_state_metrics(roles: CornerRoles, positions: torch.Tensor, tangents: torch.Tensor | None = None) -> CornerMetrics:
    """
    ``positions [B, n_out, 3]`` (device, as returned by ``DeviceProgram.solve``) and optionally
    ``tangents [B, T, n_out, 3]`` (``DeviceProgram.tangents``) -> the catalog's state metrics per state
    and, with tangents, their derivative with respect to every sweep target.
    """
    if not positions.is_cuda:
        raise RuntimeError("corner_state_metrics needs device tensors (there is no CPU fallback)")
    lib = _lib.load()
    pos = positions.to(torch.float64).contiguous()
    b, n_out = pos.shape[0], pos.shape[1]
    values = torch.empty((b, len(METRIC_NAMES)), dtype=torch.float64, device=pos.device)
    tan = deriv = None
    n_targets = 0
    if tangents is not None:
        tan = tangents.to(torch.float64).contiguous()
        if tan.shape[0] != b or tan.shape[2:] != (n_out, 3):
            raise ValueError("tangents must be [B, T, n_out, 3]")
        n_targets = tan.shape[1]
        deriv = torch.empty((b, n_targets, len(METRIC_NAMES)), dtype=torch.float64, device=pos.device)
    stream = torch.cuda.current_stream(pos.device).cuda_stream
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    with torch.cuda.device(pos.device):
        rc = lib.okx_corner_metrics_batch(C.byref(roles), b, n_out, n_targets, ptr(pos), ptr(tan), ptr(values),
                                          ptr(deriv), C.c_void_p(stream))
    _lib.check(rc, "okx_corner_metrics_batch")
    return CornerMetrics(values, deriv)


def axle_state_metrics(left: CornerRoles, right: CornerRoles, positions: torch.Tensor) -> torch.Tensor:
    """
    Axle-scope metrics (``metrics/axle_metrics.py:21-95``) of solved axle states ``positions [B, n_out, 3]``:
    ``[B, 7]`` in ``AXLE_METRIC_NAMES`` order, NaN where the reference reports None.  The per-corner rows
    of an axle are ``corner_state_metrics`` with ``axle_roles``' left / right roles.
    """
    if not positions.is_cuda:
        raise RuntimeError("axle_state_metrics needs device tensors (there is no CPU fallback)")
    lib = _lib.load()
    pos = positions.to(torch.float64).contiguous()
    values = torch.empty((pos.shape[0], len(AXLE_METRIC_NAMES)), dtype=torch.float64, device=pos.device)
    stream = torch.cuda.current_stream(pos.device).cuda_stream
    with torch.cuda.device(pos.device):
        rc = lib.okx_axle_metrics_batch(C.byref(left), C.byref(right), pos.shape[0], pos.shape[1],
                                        C.c_void_p(pos.data_ptr()), C.c_void_p(values.data_ptr()), C.c_void_p(stream))
    _lib.check(rc, "okx_axle_metrics_batch")
    return values
