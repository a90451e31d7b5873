"""
Constraint descriptors with the reference's class and attribute names (``core/constraints.py``).  They describe rows; the
solver's arithmetic lives on the device (``csrc/okx_kernels.hip`` and the generated kernels).  ``residual(positions)`` is the
reference's host-side evaluation of ONE row on a mapping of point keys to positions (``core/constraints.py:56-733``: what
its diagnostics call on a solved state), in NumPy, with the reference's definitions to the letter - ``softnorm`` where it
uses it -; it is not on the solve path.
"""

from __future__ import annotations

import copy
from typing import Any, Callable, ClassVar

import numpy as np


EPS = 1e-6        # primitives/constants.py:9 EPS_GEOMETRIC
EPS_SQ = EPS ** 2  # primitives/soft_math.py:16


def softnorm(sum_of_squares: float) -> float:
    """``sqrt(s + EPS_SQ) - EPS`` (``primitives/soft_math.py:20-27``)."""
    return float(np.sqrt(sum_of_squares + EPS_SQ) - EPS)


def _xyz(positions, key) -> np.ndarray:
    value = positions[key]
    return np.asarray(getattr(value, "data", value), dtype=np.float64)


class Constraint:
    _POINT_ATTRS: ClassVar[tuple] = ()

    def residual(self, positions) -> float:  # constraints.py:56-63
        raise NotImplementedError

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

    def residual(self, positions) -> float:  # constraints.py:125-134
        delta = _xyz(positions, self.p2) - _xyz(positions, self.p1)
        return float(softnorm(float(delta @ delta)) - self.target_distance)


class SphericalJointConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2")

    def __init__(self, p1, p2):
        self.p1, self.p2 = p1, p2

    def residual(self, positions) -> float:  # constraints.py:162-170
        delta = _xyz(positions, self.p2) - _xyz(positions, self.p1)
        return float(softnorm(float(delta @ delta)))


class AngleConstraint(Constraint):
    _POINT_ATTRS = ("v1_start", "v1_end", "v2_start", "v2_end")

    def __init__(self, v1_start, v1_end, v2_start, v2_end, target_angle: float):
        if not (0 <= target_angle <= np.pi):
            raise ValueError(f"Target angle must be in [0, pi], got {target_angle}")
        self.v1_start, self.v1_end, self.v2_start, self.v2_end = v1_start, v1_end, v2_start, v2_end
        self.target_angle = float(target_angle)

    def residual(self, positions) -> float:  # constraints.py:223-243
        v1 = _xyz(positions, self.v1_end) - _xyz(positions, self.v1_start)
        v2 = _xyz(positions, self.v2_end) - _xyz(positions, self.v2_start)
        c = np.cross(v1, v2)
        return float(np.arctan2(softnorm(float(c @ c)), float(v1 @ v2)) - self.target_angle)


class ThreePointAngleConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3")

    def __init__(self, p1, p2, p3, target_angle: float):
        if not (0 <= target_angle <= np.pi):
            raise ValueError(f"Target angle must be in [0, pi], got {target_angle}")
        self.p1, self.p2, self.p3, self.target_angle = p1, p2, p3, float(target_angle)

    def residual(self, positions) -> float:  # constraints.py:287-308
        v1 = _xyz(positions, self.p1) - _xyz(positions, self.p2)
        v2 = _xyz(positions, self.p3) - _xyz(positions, self.p2)
        c = np.cross(v1, v2)
        return float(np.arctan2(softnorm(float(c @ c)), float(v1 @ v2)) - self.target_angle)


class _TwoVectors(Constraint):
    _POINT_ATTRS = ("v1_start", "v1_end", "v2_start", "v2_end")

    def __init__(self, v1_start, v1_end, v2_start, v2_end):
        self.v1_start, self.v1_end, self.v2_start, self.v2_end = v1_start, v1_end, v2_start, v2_end


class VectorsParallelConstraint(_TwoVectors):
    def residual(self, positions) -> float:  # constraints.py:351-371
        v1 = _xyz(positions, self.v1_end) - _xyz(positions, self.v1_start)
        v2 = _xyz(positions, self.v2_end) - _xyz(positions, self.v2_start)
        c = np.cross(v1, v2)
        return float(softnorm(float(c @ c)) / (softnorm(float(v1 @ v1)) * softnorm(float(v2 @ v2))))


class VectorsPerpendicularConstraint(_TwoVectors):
    def residual(self, positions) -> float:  # constraints.py:414-429
        v1 = _xyz(positions, self.v1_end) - _xyz(positions, self.v1_start)
        v2 = _xyz(positions, self.v2_end) - _xyz(positions, self.v2_start)
        return float(float(v1 @ v2) / (softnorm(float(v1 @ v1)) * softnorm(float(v2 @ v2))))


class EqualDistanceConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3", "p4")

    def __init__(self, p1, p2, p3, p4):
        self.p1, self.p2, self.p3, self.p4 = p1, p2, p3, p4

    def residual(self, positions) -> float:  # constraints.py:466-477
        d1 = _xyz(positions, self.p2) - _xyz(positions, self.p1)
        d2 = _xyz(positions, self.p4) - _xyz(positions, self.p3)
        return float(softnorm(float(d1 @ d1)) - softnorm(float(d2 @ d2)))


class FixedAxisConstraint(Constraint):
    _POINT_ATTRS = ("point_id",)

    def __init__(self, point_id, axis, value: float):
        self.point_id, self.axis, self.value = point_id, axis, float(value)

    def residual(self, positions) -> float:  # constraints.py:508-516
        return float(_xyz(positions, self.point_id)[int(self.axis)] - self.value)


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

    def residual(self, positions) -> float:  # constraints.py:560-576
        c = np.cross(_xyz(positions, self.point_id) - self.line_point, self.line_direction)
        return float(softnorm(float(c @ c)))


class PointOnPlaneConstraint(Constraint):
    _POINT_ATTRS = ("point_id",)

    def __init__(self, point_id, plane_point, plane_normal):
        self.point_id = point_id
        self.plane_point = _vec(plane_point)
        self.plane_normal = _vec(plane_normal)

    def residual(self, positions) -> float:  # constraints.py:616-627, vector_utils/geometric.py:174-194
        return float((_xyz(positions, self.point_id) - self.plane_point) @ self.plane_normal)


class MidpointOnPlaneConstraint(Constraint):
    _POINT_ATTRS = ("point_a", "point_b")

    def __init__(self, point_a, point_b, plane_point, plane_normal):
        self.point_a, self.point_b = point_a, point_b
        self.plane_point = _vec(plane_point)
        self.plane_normal = _vec(plane_normal)

    def residual(self, positions) -> float:  # constraints.py:657-666
        a, b = _xyz(positions, self.point_a), _xyz(positions, self.point_b)
        return float(((a + (b - a) / 2.0) - self.plane_point) @ self.plane_normal)


class CoplanarPointsConstraint(Constraint):
    _POINT_ATTRS = ("p1", "p2", "p3", "p4")

    def __init__(self, p1, p2, p3, p4):
        self.p1, self.p2, self.p3, self.p4 = p1, p2, p3, p4

    def residual(self, positions) -> float:  # constraints.py:698-709: v1 . (v2 x v3), vectors from p1
        base = _xyz(positions, self.p1)
        v1, v2, v3 = (_xyz(positions, k) - base for k in (self.p2, self.p3, self.p4))
        return float(v1 @ np.cross(v2, v3))


class ScalarTripleProductConstraint(CoplanarPointsConstraint):
    def __init__(self, p1, p2, p3, p4, target_volume: float, scale: float = 1.0):
        if scale <= 0.0:
            raise ValueError(f"scale must be strictly positive, got {scale}")
        super().__init__(p1, p2, p3, p4)
        self.target_volume, self.scale = float(target_volume), float(scale)

    def residual(self, positions) -> float:  # constraints.py:731-733
        return (super().residual(positions) - self.target_volume) / self.scale
