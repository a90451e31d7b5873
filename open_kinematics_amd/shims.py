"""
Camber-shim setup solve on the device: the reference's
``solve_camber_shim_assembly`` (``suspensions/config/shims.py:284-501``) and
``DoubleWishboneSuspension.apply_camber_shim`` (``corner/double_wishbone.py:501-570``) for a whole
batch of geometries / shim stacks in one launch (``okx_camber_shim_batch``).  The point table is
rewritten in place in HBM and feeds ``DeviceProgram.rebind`` directly; there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .enums import PointID

P = PointID
MAX_POINTS = 8      # OKX_SHIM_MAX_POINTS
SHIM_PARAMS = 11    # OKX_SHIM_PARAMS


class ShimRoles(C.Structure):
    """ctypes mirror of ``okx_shim_roles``."""

    _fields_ = [
        ("upper_outboard", C.c_int32), ("lower_outboard", C.c_int32),
        ("upper_inboard_front", C.c_int32), ("upper_inboard_rear", C.c_int32),
        ("heading_inboard", C.c_int32), ("heading_outboard", C.c_int32),
        ("n_upright_points", C.c_int32), ("upright_point", C.c_int32 * MAX_POINTS),
        ("rocker", C.c_int32),
        ("rocker_axis_a", C.c_int32), ("rocker_axis_b", C.c_int32),
        ("pushrod_inboard", C.c_int32), ("pushrod_outboard", C.c_int32),
        ("n_rocker_points", C.c_int32), ("rocker_point", C.c_int32 * MAX_POINTS),
    ]


SHIM_INFO_DTYPE = np.dtype([
    ("residual_norm", "<f8"), ("max_residual", "<f8"), ("upright_angle_rad", "<f8"), ("rocker_angle_rad", "<f8"),
    ("wishbone_angle_rad", "<f8"), ("converged", "<i4"), ("iterations", "<i4"),
])  # okx_shim_info


@dataclass(frozen=True)
class CamberShimConfig:
    """``schema/config.py:52-70``: the split upright's shim face (two datum points, unit normal) and thicknesses."""

    shim_face_point_a: tuple
    shim_face_point_b: tuple
    shim_face_normal: tuple
    design_thickness: float
    setup_thickness: float

    def __post_init__(self) -> None:
        a, b = np.asarray(self.shim_face_point_a, float), np.asarray(self.shim_face_point_b, float)
        if float(np.linalg.norm(b - a)) < 1e-6:
            raise ValueError("shim_face_point_a and shim_face_point_b must be distinct")

    @property
    def unchanged(self) -> bool:
        """The reference's early exit (``shims.py:346-357``): equal thicknesses, nothing moves."""
        return abs(self.setup_thickness - self.design_thickness) < 1e-6

    def row(self, setup_thickness: float | None = None) -> np.ndarray:
        """The ``OKX_SHIM_PARAMS`` numbers of one geometry."""
        normal = np.asarray(self.shim_face_normal, float)
        normal = normal / float(np.linalg.norm(normal))  # Direction3 normalises on construction
        t = self.setup_thickness if setup_thickness is None else setup_thickness
        return np.concatenate([np.asarray(self.shim_face_point_a, float), np.asarray(self.shim_face_point_b, float),
                               normal, [float(self.design_thickness), float(t)]])

    def mirrored(self) -> "CamberShimConfig":
        """Reflection through the vehicle XZ plane (``suspensions/build.py:357-375``)."""
        flip = lambda v: (float(v[0]), -float(v[1]), float(v[2]))  # noqa: E731
        return CamberShimConfig(flip(self.shim_face_point_a), flip(self.shim_face_point_b), flip(self.shim_face_normal),
                                self.design_thickness, self.setup_thickness)


def make_shim_roles(index, heading_inboard, heading_outboard, upright_points, rocker_points=None) -> ShimRoles:
    """
    ``index``: point key -> row of the point table.  ``upright_points``: what the upright carries
    (``upright_attachment_points()``); ``rocker_points``: the rocker group when the pushrod is upright-mounted
    (``rotate_rocker_group``), else None.
    """
    upright = [index(k) for k in upright_points]
    rocker = [index(k) for k in (rocker_points or ())]
    if len(upright) > MAX_POINTS or len(rocker) > MAX_POINTS:
        raise ValueError(f"at most {MAX_POINTS} upright / rocker points")
    pad = lambda v: (C.c_int32 * MAX_POINTS)(*(v + [-1] * (MAX_POINTS - len(v))))  # noqa: E731
    coupled = rocker_points is not None
    return ShimRoles(
        upper_outboard=index(P.UPPER_WISHBONE_OUTBOARD), lower_outboard=index(P.LOWER_WISHBONE_OUTBOARD),
        upper_inboard_front=index(P.UPPER_WISHBONE_INBOARD_FRONT), upper_inboard_rear=index(P.UPPER_WISHBONE_INBOARD_REAR),
        heading_inboard=index(heading_inboard), heading_outboard=index(heading_outboard),
        n_upright_points=len(upright), upright_point=pad(upright), rocker=int(coupled),
        rocker_axis_a=index(P.ROCKER_AXIS_A) if coupled else -1, rocker_axis_b=index(P.ROCKER_AXIS_B) if coupled else -1,
        pushrod_inboard=index(P.PUSHROD_INBOARD) if coupled else -1,
        pushrod_outboard=index(P.PUSHROD_OUTBOARD) if coupled else -1,
        n_rocker_points=len(rocker), rocker_point=pad(rocker),
    )


def shim_roles(suspension, point_keys) -> ShimRoles:
    """Roles of a double-wishbone corner inside a point table whose rows are ``point_keys``."""
    keys = list(point_keys)

    def index(key) -> int:
        try:
            return keys.index(key)
        except ValueError:
            raise ValueError(f"shim role point {key!r} is not in the point table") from None

    return make_shim_roles(index, suspension.heading_link.inboard_point, suspension.heading_link.outboard_point,
                           suspension.upright_attachment_points(), suspension.shim_rocker_points())


def camber_shim_setup(roles: ShimRoles, points: torch.Tensor, shim: torch.Tensor, check: bool = True):
    """
    ``points [G, P, 3]`` (device, authored positions; rewritten IN PLACE to the setup pose) and
    ``shim [G, 11]`` (``CamberShimConfig.row``) -> ``(points, info)`` with ``info`` a numpy record array
    (``SHIM_INFO_DTYPE``) when ``check`` is set — a non-converged assembly then raises like the reference
    (``shims.py:451-462``) — or the raw device bytes otherwise.
    """
    if not points.is_cuda or not shim.is_cuda:
        raise RuntimeError("camber_shim_setup needs device tensors (there is no CPU fallback)")
    if points.dtype != torch.float64 or not points.is_contiguous() or points.dim() != 3 or points.shape[2] != 3:
        raise ValueError("points must be a contiguous float64 [G, P, 3] tensor")
    shim = shim.to(torch.float64).contiguous()
    if shim.shape != (points.shape[0], SHIM_PARAMS):
        raise ValueError(f"shim must be [G, {SHIM_PARAMS}]")
    lib = _lib.load()
    info = torch.empty((points.shape[0], SHIM_INFO_DTYPE.itemsize), dtype=torch.uint8, device=points.device)
    stream = torch.cuda.current_stream(points.device).cuda_stream
    with torch.cuda.device(points.device):
        rc = lib.okx_camber_shim_batch(C.byref(roles), points.shape[0], points.shape[1], C.c_void_p(points.data_ptr()),
                                       C.c_void_p(shim.data_ptr()), C.c_void_p(info.data_ptr()), C.c_void_p(stream))
    _lib.check(rc, "okx_camber_shim_batch")
    if not check:
        return points, info
    records = info.cpu().numpy().view(SHIM_INFO_DTYPE).reshape(-1)
    bad = np.flatnonzero(records["converged"] == 0)
    if bad.size:
        worst = float(records["max_residual"][bad].max())
        raise RuntimeError(f"Camber shim assembly solve did not satisfy its constraints for {bad.size} of "
                           f"{records.size} geometries: maximum residual {worst:.6g} exceeds tolerance 0.001.")
    return points, records


def apply_camber_shim(suspension, positions: dict, device: str = "cuda:0") -> None:
    """
    ``DoubleWishboneSuspension.apply_camber_shim`` for ONE corner at load time: moves the entries of
    ``positions`` (PointID -> Point3) to the setup pose.  Runs the device kernel on a one-row table.
    """
    shim = suspension.camber_shim
    if shim is None or shim.unchanged:
        return
    if not torch.cuda.is_available():
        raise RuntimeError("a camber-shim setup solve needs the GPU (okx_camber_shim_batch); there is no CPU fallback")
    keys = list(positions)
    table = torch.as_tensor(np.asarray([positions[k].data for k in keys], dtype=np.float64)[None], device=device).contiguous()
    row = torch.as_tensor(shim.row()[None], device=device)
    camber_shim_setup(shim_roles(suspension, keys), table, row)
    moved = table[0].cpu().numpy()
    for k, xyz in zip(keys, moved):
        positions[k].data[:] = xyz
