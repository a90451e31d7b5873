"""
CPU restatement (numpy + a 30-line forward-mode scalar) of the reference's corner state metrics and
their directional derivatives.  TEST INFRASTRUCTURE ONLY.

Follows ``kinematics/core/metrics/angles.py:22-132`` (camber, caster, KPI, toe = roadwheel angle),
``travel.py:19-45`` (wheel travel, half-track), ``steering_geometry.py:22-76`` and
``context.py:82-138`` (wheel axis, steering axis, steering-axis / ground-plane intersection).  The
derivative of a metric along a tangent field is what ``metrics/derivatives.py`` evaluates with the
reference's dual numbers (``primitives/dual.py``).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

METRIC_NAMES = ("camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius",
                "mechanical_trail")


@dataclass
class D:
    """value + derivative along one direction."""

    v: float
    d: float = 0.0

    def __add__(self, o):
        o = _lift(o)
        return D(self.v + o.v, self.d + o.d)

    __radd__ = __add__

    def __sub__(self, o):
        o = _lift(o)
        return D(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        return _lift(o) - self

    def __neg__(self):
        return D(-self.v, -self.d)

    def __mul__(self, o):
        o = _lift(o)
        return D(self.v * o.v, self.v * o.d + self.d * o.v)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = _lift(o)
        q = self.v / o.v
        return D(q, (self.d - q * o.d) / o.v)


def _lift(x) -> D:
    return x if isinstance(x, D) else D(float(x), 0.0)


def atan2(y: D, x: D) -> D:
    return D(math.atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v))


def sqrt(a: D) -> D:
    r = math.sqrt(a.v)
    return D(r, a.d / (2.0 * r))


def corner_metrics(pos: dict, vel: dict | None, side: float, design_wheel_center_z: float):
    """
    ``pos`` / ``vel``: role name -> xyz for wheel_center, contact_patch, axle_inboard, axle_outboard,
    steer_lower, steer_upper.  Returns ``(values[8], derivatives[8])``.
    """
    def point(name):
        v = vel[name] if vel is not None else (0.0, 0.0, 0.0)
        return [D(float(pos[name][k]), float(v[k])) for k in range(3)]

    wc, cp = point("wheel_center"), point("contact_patch")
    axi, axo = point("axle_inboard"), point("axle_outboard")
    lower, upper = point("steer_lower"), point("steer_upper")
    axle = [axo[k] - axi[k] for k in range(3)]
    steer = [upper[k] - lower[k] for k in range(3)]
    deg = 180.0 / math.pi
    # angles.py:22-50: wheel_up = axle x X * -side
    up_y, up_z = -side * axle[2], side * axle[1]
    angle = atan2(up_y, up_z)
    camber = deg * (angle if side > 0 else -angle)
    caster = deg * atan2(-steer[0], steer[2])
    kpi = deg * atan2(-side * steer[1], steer[2])
    toe = deg * (atan2(axle[0], axle[1]) if side > 0 else atan2(axle[0], -axle[1]))
    travel = wc[2] - design_wheel_center_z
    half_track = cp[1] if cp[1].v >= 0 else -cp[1]
    t = (cp[2] - lower[2]) / steer[2]  # context.py:119-138
    gx, gy = lower[0] + t * steer[0], lower[1] + t * steer[1]
    an = sqrt(axle[0] * axle[0] + axle[1] * axle[1])
    scrub = -(((gx - cp[0]) * axle[0] + (gy - cp[1]) * axle[1]) / an)
    trail = gx - cp[0]
    out = [camber, caster, kpi, toe, travel, half_track, scrub, trail]
    return np.array([o.v for o in out]), np.array([o.d for o in out])
