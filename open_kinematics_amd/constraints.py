"""
Constraint descriptors with the reference's class and attribute names
(``core/constraints.py``).  They describe rows; the arithmetic lives on the device
(``csrc/okx_kernels.hip``), so there is no host ``residual()`` here.
"""

from __future__ import annotations

import copy
from typing import Any, Callable, ClassVar

import numpy as np


class Constraint:
    _POINT_ATTRS: ClassVar[tuple] = ()

    @property
    def involved_points(self) -> set:
        return {getattr(self, a) for a in self._POINT_ATTRS}

    def remap(self, mapping: Callable[[Any], Any]) -> "Constraint":
        """Same constraint in another point namespace (``constraints.py:65-86``)."""
        new = copy.copy(self)
        for attr in self._POINT_ATTRS:
            setattr(new, attr, mapping(getattr(self, attr)))
        return new

    def __repr__(self) -> str:
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in vars(self).items())})"


class DistanceConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2")

    def __init__(self, p1, p2, target_distance: float):
        if target_distance < 0:
            raise ValueError(f"Target distance must be non-negative, got {target_distance}")
        self.p1, self.p2, self.target_distance = p1, p2, float(target_distance)


class SphericalJointConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2")

    def __init__(self, p1, p2):
        self.p1, self.p2 = p1, p2


class AngleConstraint(Constraint):
    _POINT_ATTRS = ("v1_start", "v1_end", "v2_start", "v2_end")

    def __init__(self, v1_start, v1_end, v2_start, v2_end, target_angle: float):
        if not (0 <= target_angle <= np.pi):
            raise ValueError(f"Target angle must be in [0, pi], got {target_angle}")
        self.v1_start, self.v1_end, self.v2_start, self.v2_end = v1_start, v1_end, v2_start, v2_end
        self.target_angle = float(target_angle)


class ThreePointAngleConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3")

    def __init__(self, p1, p2, p3, target_angle: float):
        if not (0 <= target_angle <= np.pi):
            raise ValueError(f"Target angle must be in [0, pi], got {target_angle}")
        self.p1, self.p2, self.p3, self.target_angle = p1, p2, p3, float(target_angle)


class _TwoVectors(Constraint):
    _POINT_ATTRS = ("v1_start", "v1_end", "v2_start", "v2_end")

    def __init__(self, v1_start, v1_end, v2_start, v2_end):
        self.v1_start, self.v1_end, self.v2_start, self.v2_end = v1_start, v1_end, v2_start, v2_end


class VectorsParallelConstraint(_TwoVectors):
    pass


class VectorsPerpendicularConstraint(_TwoVectors):
    pass


class EqualDistanceConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3", "p4")

    def __init__(self, p1, p2, p3, p4):
        self.p1, self.p2, self.p3, self.p4 = p1, p2, p3, p4


class FixedAxisConstraint(Constraint):
    _POINT_ATTRS = ("point_id",)

    def __init__(self, point_id, axis, value: float):
        self.point_id, self.axis, self.value = point_id, axis, float(value)


def _vec(v) -> np.ndarray:
    return np.array(getattr(v, "data", v), dtype=np.float64)


class PointOnLineConstraint(Constraint):
    _POINT_ATTRS = ("point_id",)

    def __init__(self, point_id, line_point, line_direction):
        self.point_id = point_id
        self.line_point = _vec(line_point)
        direction = _vec(line_direction)
        norm = float(np.linalg.norm(direction))
        if norm < 1e-6:
            raise ValueError("line_direction has zero length")
        self.line_direction = direction / norm


class PointOnPlaneConstraint(Constraint):
    _POINT_ATTRS = ("point_id",)

    def __init__(self, point_id, plane_point, plane_normal):
        self.point_id = point_id
        self.plane_point = _vec(plane_point)
        self.plane_normal = _vec(plane_normal)


class MidpointOnPlaneConstraint(Constraint):
    _POINT_ATTRS = ("point_a", "point_b")

    def __init__(self, point_a, point_b, plane_point, plane_normal):
        self.point_a, self.point_b = point_a, point_b
        self.plane_point = _vec(plane_point)
        self.plane_normal = _vec(plane_normal)


class CoplanarPointsConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3", "p4")

    def __init__(self, p1, p2, p3, p4):
        self.p1, self.p2, self.p3, self.p4 = p1, p2, p3, p4


class ScalarTripleProductConstraint(CoplanarPointsConstraint):
    def __init__(self, p1, p2, p3, p4, target_volume: float, scale: float = 1.0):
        if scale <= 0.0:
            raise ValueError(f"scale must be strictly positive, got {scale}")
        super().__init__(p1, p2, p3, p4)
        self.target_volume, self.scale = float(target_volume), float(scale)
