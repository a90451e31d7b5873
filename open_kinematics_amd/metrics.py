"""
Corner state metrics of solved sweep states on the device: camber, caster, KPI,
roadwheel angle, wheel travel, half-track, scrub radius and mechanical trail — the reference's
``core/metrics/angles.py``, ``travel.py`` and ``steering_geometry.py`` — and their derivatives along
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
from .enums import PointID

METRIC_NAMES = ("camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius",
                "mechanical_trail")  # column order = OKX_METRIC_* (include/okx.h); names = metrics/catalog.py


class CornerRoles(C.Structure):
    """ctypes mirror of ``okx_corner_roles``."""

    _fields_ = [
        ("wheel_center", C.c_int32), ("contact_patch", C.c_int32),
        ("axle_inboard", C.c_int32), ("axle_outboard", C.c_int32),
        ("steer_lower", C.c_int32), ("steer_upper", C.c_int32),
        ("side_sign", C.c_double), ("design_wheel_center_z", C.c_double),
    ]


def corner_roles(suspension, program) -> CornerRoles:
    """
    Role indices into ``program.out_point`` from the corner's role hooks
    (``wheel_axis_points()``, ``steering_axis_points()``, ``side``; reference
    ``suspensions/corner/base.py``, ``metrics/context.py:82-104``).
    """
    out_keys = [program.point_keys[k] for k in program.out_point]

    def index(key) -> int:
        try:
            return out_keys.index(key)
        except ValueError:
            raise ValueError(f"metric role point {key!r} is not among the program's output points") from None

    axle_in, axle_out = suspension.wheel_axis_points()
    lower, upper = suspension.steering_axis_points()
    side = getattr(suspension, "lateral_sign", None)
    if side is None:
        side = suspension.side.lateral_sign
    design = suspension.initial_state().positions[PointID.WHEEL_CENTER]
    return CornerRoles(
        wheel_center=index(PointID.WHEEL_CENTER), contact_patch=index(PointID.CONTACT_PATCH_CENTER),
        axle_inboard=index(axle_in), axle_outboard=index(axle_out),
        steer_lower=index(lower), steer_upper=index(upper),
        side_sign=float(side), design_wheel_center_z=float(np.asarray(getattr(design, "data", design))[2]),
    )


@dataclass
class CornerMetrics:
    values: torch.Tensor              # [B, 8] float64, device
    derivatives: torch.Tensor | None  # [B, T, 8]: d metric / d target (deg or mm per mm of target)

    def column(self, name: str) -> torch.Tensor:
        return self.values[:, METRIC_NAMES.index(name)]

    def derivative(self, name: str, target_index: int) -> torch.Tensor:
        if self.derivatives is None:
            raise ValueError("no tangents were given")
        return self.derivatives[:, target_index, METRIC_NAMES.index(name)]


def corner_state_metrics(roles: CornerRoles, positions: torch.Tensor, tangents: torch.Tensor | None = None) -> CornerMetrics:
    """
    ``positions [B, n_out, 3]`` (device, as returned by ``DeviceProgram.solve``) and optionally
    ``tangents [B, T, n_out, 3]`` (``DeviceProgram.tangents``) -> the eight state metrics per state
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
