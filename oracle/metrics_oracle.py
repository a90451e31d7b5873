"""
CPU restatement (numpy + a 30-line forward-mode scalar) of the reference's corner state metrics and
their directional derivatives.  TEST INFRASTRUCTURE ONLY.

Follows ``kinematics/core/metrics/angles.py:22-132`` (camber, caster, KPI, toe = roadwheel angle),
``travel.py:19-45`` (wheel travel, half-track), ``steering_geometry.py:22-76`` and
``context.py:82-138`` (wheel axis, steering axis, steering-axis / ground-plane intersection),
``swing_arms.py:45-88``, ``anti_geometry.py:32-206``, ``travel.py:48-62`` (damper length), the instant axis of
``corner/double_wishbone.py:352-430`` / ``corner/macpherson.py:325-378`` over
``vector_utils/geometric.py:216-352``, and ``axle_metrics.py:21-95`` (values only).  The
derivative of a metric along a tangent field is what ``metrics/derivatives.py`` evaluates with the
reference's dual numbers (``primitives/dual.py``).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

METRIC_NAMES = (
    "camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius", "mechanical_trail",
    "svic_x", "svic_z", "svsa_length", "fvic_y", "fvic_z", "fvsa_length", "damper_length", "svsa_angle",
    "anti_dive", "anti_lift", "anti_squat",
)
AXLE_METRIC_NAMES = ("heave", "roll", "ride_height_change", "track", "roll_center_y", "roll_center_z",
                     "rack_displacement")
EPS_GEOMETRIC = 1e-6  # primitives/constants.py:9


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
    steer_lower, steer_upper (+ optionally damper_top, damper_bottom).  Returns ``(values[19], derivatives[19])``
    with the first eight entries (and damper_length when its points are given) filled, NaN elsewhere;
    ``geometry_metrics`` below supplies the values of the rest.
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
    values, derivs = np.full(len(METRIC_NAMES), np.nan), np.full(len(METRIC_NAMES), np.nan)
    values[:8], derivs[:8] = [o.v for o in out], [o.d for o in out]
    if "damper_top" in pos:  # travel.py:48-62 (the reference's deriv_damper_length_wrt_hub_z differentiates this)
        top, bottom = point("damper_top"), point("damper_bottom")
        strut = [top[k] - bottom[k] for k in range(3)]
        length = sqrt(strut[0] * strut[0] + strut[1] * strut[1] + strut[2] * strut[2])
        k = METRIC_NAMES.index("damper_length")
        values[k], derivs[k] = length.v, length.d
    return values, derivs


# ---- instant centres, swing arms, anti-geometry, damper length, axle metrics (values only) ----


def _plane(a, b, c):
    """geometric.py:216-252."""
    normal = np.cross(b - a, c - a)
    mag = float(np.linalg.norm(normal))
    if mag < EPS_GEOMETRIC:
        return None
    n = normal / mag
    return n, -float(np.dot(n, a))


def _two_planes(n1, d1, n2, d2):
    """geometric.py:255-290."""
    direction = np.cross(n1, n2)
    m2 = float(np.dot(direction, direction))
    if m2 < EPS_GEOMETRIC * EPS_GEOMETRIC:
        return None
    point = np.cross(d2 * n1 - d1 * n2, direction) / m2
    return point, direction / math.sqrt(m2)


def _at_coordinate(point, direction, axis: int, value: float):
    """geometric.py:316-352."""
    if abs(direction[axis]) < EPS_GEOMETRIC:
        return None
    return point + (value - point[axis]) / direction[axis] * direction


def instant_axis(kind: str | None, pts):
    """``compute_instant_axis``: ``two_planes`` (double wishbone) or ``plane_and_strut`` (MacPherson)."""
    pts = [np.asarray(p, dtype=np.float64) for p in pts]
    if kind == "two_planes":
        upper, lower = _plane(*pts[0:3]), _plane(*pts[3:6])
        if upper is None or lower is None:
            return None
        return _two_planes(*upper, *lower)
    if kind == "plane_and_strut":
        arm = _plane(*pts[0:3])
        if arm is None:
            return None
        strut = pts[3] - pts[2]
        axis = strut / float(np.linalg.norm(strut))
        return _two_planes(*arm, axis, -float(np.dot(axis, pts[3])))
    return None


def instant_centres(kind, pts, wheel_center):
    """(SVIC, FVIC), each a point or None."""
    line = instant_axis(kind, pts)
    if line is None:
        return None, None
    return _at_coordinate(*line, 1, float(wheel_center[1])), _at_coordinate(*line, 0, float(wheel_center[0]))


def geometry_metrics(pos: dict, side: float, axis_kind, axis_points, damper=None, wheelbase=float("nan"),
                     cg_z=float("nan"), front_brake_bias=None, axle_position=None, driven_axle=None) -> np.ndarray:
    """The eleven catalog entries after mechanical trail (METRIC_NAMES[8:]); None -> NaN."""
    nan = float("nan")
    wc, cp = np.asarray(pos["wheel_center"], float), np.asarray(pos["contact_patch"], float)
    svic, fvic = instant_centres(axis_kind, axis_points, wc)
    out = dict.fromkeys(METRIC_NAMES[8:], nan)
    if damper is not None:
        out["damper_length"] = float(np.linalg.norm(np.asarray(damper[0], float) - np.asarray(damper[1], float)))
    if fvic is not None:
        out["fvic_y"], out["fvic_z"] = float(fvic[1]), float(fvic[2])
        dy, dz = float(fvic[1] - cp[1]), float(fvic[2] - cp[2])
        out["fvsa_length"] = float(math.sqrt(dy * dy + dz * dz) * (-side * np.sign(dy)))
    if svic is not None:
        out["svic_x"], out["svic_z"] = float(svic[0]), float(svic[2])
        out["svsa_length"] = float(svic[0] - cp[0])
        run, rise = float(svic[0]) - float(cp[0]), float(svic[2]) - float(cp[2])
        height = cg_z - float(cp[2])
        height_ok = height > EPS_GEOMETRIC
        if abs(run) >= EPS_GEOMETRIC:
            out["svsa_angle"] = math.degrees(math.atan(rise / run))
            if height_ok and front_brake_bias is not None and axle_position == "front":
                out["anti_dive"] = 100.0 * front_brake_bias * (wheelbase / height) * (rise / (float(cp[0]) - float(svic[0])))
            if height_ok and front_brake_bias is not None and axle_position == "rear":
                out["anti_lift"] = 100.0 * (1.0 - front_brake_bias) * (wheelbase / height) * (rise / run)
        if driven_axle is not None and axle_position is not None and driven_axle == axle_position:
            drive_run = float(wc[0]) - float(svic[0]) if axle_position == "front" else float(svic[0]) - float(wc[0])
            if abs(drive_run) >= EPS_GEOMETRIC and height_ok:
                out["anti_squat"] = 100.0 * (wheelbase / height) * ((float(svic[2]) - float(wc[2])) / drive_run)
    return np.array([out[k] for k in METRIC_NAMES[8:]])


def axle_metrics(sides: dict) -> np.ndarray:
    """
    ``sides['left'|'right']`` = dict(wheel_center, contact_patch, design_wheel_center_z, design_contact_patch_z,
    axis_kind, axis_points, rack_y (or None), design_rack_y).  -> AXLE_METRIC_NAMES order.
    """
    nan = float("nan")
    wdz, cdz, cy, lines = {}, {}, {}, []
    for name in ("left", "right"):
        s = sides[name]
        wc, cp = np.asarray(s["wheel_center"], float), np.asarray(s["contact_patch"], float)
        wdz[name] = float(wc[2]) - s["design_wheel_center_z"]
        cdz[name] = float(cp[2]) - s["design_contact_patch_z"]
        cy[name] = float(cp[1])
        _, fvic = instant_centres(s["axis_kind"], s["axis_points"], wc)
        lines.append(None if fvic is None else (float(cp[1]), float(cp[2]), float(fvic[1] - cp[1]), float(fvic[2] - cp[2])))
    track = abs(cy["left"] - cy["right"])
    rcy = rcz = nan
    if lines[0] is not None and lines[1] is not None:
        left, right = lines
        den = left[2] * right[3] - left[3] * right[2]
        if abs(den) >= EPS_GEOMETRIC:
            t = ((right[0] - left[0]) * right[3] - (right[1] - left[1]) * right[2]) / den
            rcy, rcz = left[0] + t * left[2], left[1] + t * left[3]
    rack = sides["left"].get("rack_y")
    return np.array([
        0.5 * (wdz["left"] + wdz["right"]),
        math.degrees(math.atan2(wdz["left"] - wdz["right"], track)),
        -0.5 * (cdz["left"] + cdz["right"]),
        track, rcy, rcz,
        nan if rack is None else float(rack) - sides["left"]["design_rack_y"],
    ])


def rotation_about_fixed_axis_deg(current, velocity, design, axis_point, axis_dir, scale: float = 1.0):
    """
    ``metrics/kernels.py:58-76`` (value, derivative along ``velocity``): signed rotation of ``current`` from
    ``design`` about the fixed axis, degrees, times ``scale`` (``Side.lateral_sign`` for the rocker angle).
    """
    v = velocity if velocity is not None else (0.0, 0.0, 0.0)
    cur = [D(float(current[k]) - float(axis_point[k]), float(v[k])) for k in range(3)]
    dr = [float(design[k]) - float(axis_point[k]) for k in range(3)]
    a = [float(x) for x in axis_dir]
    dot = lambda p, q: p[0] * q[0] + p[1] * q[1] + p[2] * q[2]  # noqa: E731
    dd, cd = dot(dr, a), dot(cur, a)
    dperp = [dr[k] - dd * a[k] for k in range(3)]
    cperp = [cur[k] - cd * a[k] for k in range(3)]
    cross = [dr[1] * cur[2] - dr[2] * cur[1], dr[2] * cur[0] - dr[0] * cur[2], dr[0] * cur[1] - dr[1] * cur[0]]
    angle = atan2(_lift(dot(a, cross)), _lift(dot(dperp, cperp)))
    return scale * math.degrees(1.0) * angle.v, scale * math.degrees(1.0) * angle.d
