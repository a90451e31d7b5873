"""
Derived points on the host (reference ``core/points/derived/``): only used to complete the
DESIGN state (wheel centre, contact patch, strut clamp ...) when a geometry is loaded; during
a solve they are evaluated on the device.  Function names and ``functools.partial`` keywords
are the contract ``program.flatten_problem`` recognises.
"""

from __future__ import annotations

from dataclasses import dataclass
from functools import partial
from typing import Any, Mapping

import numpy as np

from .enums import PointID
from .program import derived_update_order
from .state import Point3


def _a(p) -> np.ndarray:
    return np.asarray(getattr(p, "data", p), dtype=np.float64)


def _unit(v: np.ndarray) -> np.ndarray:
    norm = float(np.linalg.norm(v))
    if norm < 1e-6:
        raise ValueError("Cannot normalize zero-length vector")
    return v / norm


def get_point_along_line(positions, start_point, end_point, distance_from_start):
    start = _a(positions[start_point])
    return Point3(start + _unit(_a(positions[end_point]) - start) * distance_from_start)


def get_axle_midpoint(positions):
    p1, p2 = _a(positions[PointID.AXLE_INBOARD]), _a(positions[PointID.AXLE_OUTBOARD])
    return Point3(p1 + (p2 - p1) / 2)


def get_wheel_center(positions, wheel_offset):
    p1, p2 = _a(positions[PointID.AXLE_OUTBOARD]), _a(positions[PointID.AXLE_INBOARD])
    return Point3(p1 - _unit(p1 - p2) * wheel_offset)


def get_wheel_inboard(positions, wheel_width):
    p1, p2 = _a(positions[PointID.AXLE_INBOARD]), _a(positions[PointID.WHEEL_CENTER])
    return Point3(p2 - _unit(p2 - p1) * (wheel_width / 2))


def get_wheel_outboard(positions, wheel_width):
    p1, p2 = _a(positions[PointID.WHEEL_CENTER]), _a(positions[PointID.AXLE_INBOARD])
    return Point3(p1 + _unit(p1 - p2) * (wheel_width / 2))


def get_contact_patch_center(positions, tire_radius):
    wc = _a(positions[PointID.WHEEL_CENTER])
    axle = _unit(_a(positions[PointID.AXLE_OUTBOARD]) - _a(positions[PointID.AXLE_INBOARD]))
    down = -1 * np.array([0.0, 0.0, 1.0])
    wheel_down = _unit(down - np.dot(down, axle) * axle)
    return Point3(wc + wheel_down * tire_radius)


@dataclass(frozen=True)
class DerivedPointsSpec:
    functions: Mapping[Any, Any]
    dependencies: Mapping[Any, set]

    def __post_init__(self):
        if set(self.functions) != set(self.dependencies):
            raise ValueError("derived functions and dependencies must declare the same points")

    def all_points(self) -> set:
        return set(self.functions.keys())


class DerivedPointsManager:
    """Topological evaluation order (``manager.py:89-197``)."""

    def __init__(self, spec: DerivedPointsSpec):
        self.spec = spec
        self.update_order = derived_update_order(spec)

    def update_in_place(self, positions: dict) -> None:
        for key in self.update_order:
            positions[key] = self.spec.functions[key](positions)


def build_wheel_derived_spec(wheel_offset: float, section_width: float, tire_radius: float) -> DerivedPointsSpec:
    """Standard wheel points from the axle pair (``definitions.py:183-216``)."""
    P = PointID
    functions = {
        P.AXLE_MIDPOINT: get_axle_midpoint,
        P.WHEEL_CENTER: partial(get_wheel_center, wheel_offset=wheel_offset),
        P.WHEEL_INBOARD: partial(get_wheel_inboard, wheel_width=section_width),
        P.WHEEL_OUTBOARD: partial(get_wheel_outboard, wheel_width=section_width),
        P.CONTACT_PATCH_CENTER: partial(get_contact_patch_center, tire_radius=tire_radius),
    }
    dependencies = {
        P.AXLE_MIDPOINT: {P.AXLE_INBOARD, P.AXLE_OUTBOARD},
        P.WHEEL_CENTER: {P.AXLE_INBOARD, P.AXLE_OUTBOARD},
        P.WHEEL_INBOARD: {P.WHEEL_CENTER, P.AXLE_INBOARD},
        P.WHEEL_OUTBOARD: {P.WHEEL_CENTER, P.AXLE_INBOARD},
        P.CONTACT_PATCH_CENTER: {P.WHEEL_CENTER, P.AXLE_INBOARD, P.AXLE_OUTBOARD},
    }
    return DerivedPointsSpec(functions, dependencies)
