"""
CPU restatement (numpy + scipy's MINPACK ``lm``) of the reference's camber-shim setup solve.
TEST INFRASTRUCTURE ONLY.

Follows ``kinematics/core/suspensions/config/shims.py``: the residuals of ``:118-268`` (datum A / B
closure, face-normal alignment, heading-link length, optional upright-mounted pushrod length), the
context of ``:339-440`` (design offsets and invariant lengths), the solve of ``:442-470``
(``solve_least_squares_problem``: scipy ``least_squares(method="lm")`` with ftol 1e-5, xtol 1e-9,
gtol 1e-9, finite-difference Jacobian) and the pose write-back of
``corner/double_wishbone.py:501-570`` (UBJ on its arc, upright attachments about the LBJ, rocker group
about its axis).  Rotations follow ``vector_utils/geometric.py:351-400``.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.optimize import least_squares

EPS_NUMERICAL = 1e-15
EPS_GEOMETRIC = 1e-6


def rotate(v: np.ndarray, rotvec: np.ndarray) -> np.ndarray:
    """geometric.py:351-376 (Rodrigues with a rotation vector; identity below 1e-15 rad)."""
    angle = float(np.linalg.norm(rotvec))
    if angle < EPS_NUMERICAL:
        return v.copy()
    k = rotvec / angle
    return v * np.cos(angle) + np.cross(k, v) * np.sin(angle) + k * (np.dot(k, v) * (1.0 - np.cos(angle)))


def rotate_about_axis(point, pivot, axis, angle):
    """geometric.py:377-400."""
    v = point - pivot
    return pivot + v * np.cos(angle) + np.cross(axis, v) * np.sin(angle) + axis * (np.dot(axis, v) * (1.0 - np.cos(angle)))


@dataclass
class ShimConfig:
    point_a: np.ndarray
    point_b: np.ndarray
    normal: np.ndarray
    design_thickness: float
    setup_thickness: float


@dataclass
class Context:
    """shims.py:97-116 (+ :58-66 for the rocker part)."""

    t: float
    n0: np.ndarray
    axis: np.ndarray
    hl_in: np.ndarray
    hl_len: float
    lbj: np.ndarray
    front: np.ndarray
    front_to_ubj: np.ndarray
    ua: np.ndarray
    ub: np.ndarray
    la: np.ndarray
    lb: np.ndarray
    l_hl: np.ndarray
    rocker: tuple | None = None  # (axis point, axis direction, axis -> pushrod inboard, lbj -> pushrod outboard, length)


def build_context(pos: dict, shim: ShimConfig, hl_in: str, hl_out: str, rocker: dict | None) -> Context:
    """shims.py:339-440.  ``pos``: name -> xyz for ubj, lbj, uwb_front, uwb_rear, the heading link, rocker points."""
    ubj, lbj, front, rear = (np.asarray(pos[k], float) for k in ("ubj", "lbj", "uwb_front", "uwb_rear"))
    half = 0.5 * shim.design_thickness
    n0 = np.asarray(shim.normal, float)
    ca, cb = shim.point_a - half * n0, shim.point_b - half * n0
    ba, bb = shim.point_a + half * n0, shim.point_b + half * n0
    axis = (rear - front) / float(np.linalg.norm(rear - front))
    inboard, outboard = np.asarray(pos[hl_in], float), np.asarray(pos[hl_out], float)
    rk = None
    if rocker is not None:
        a, b = np.asarray(pos[rocker["axis_a"]], float), np.asarray(pos[rocker["axis_b"]], float)
        pi, po = np.asarray(pos[rocker["pushrod_inboard"]], float), np.asarray(pos[rocker["pushrod_outboard"]], float)
        rk = (a, (b - a) / float(np.linalg.norm(b - a)), pi - a, po - lbj, float(np.linalg.norm(po - pi)))
    return Context(
        t=float(shim.setup_thickness), n0=n0, axis=axis, hl_in=inboard, hl_len=float(np.linalg.norm(outboard - inboard)),
        lbj=lbj, front=front, front_to_ubj=ubj - front, ua=ca - ubj, ub=cb - ubj, la=ba - lbj, lb=bb - lbj,
        l_hl=outboard - lbj, rocker=rk)


def residuals(x: np.ndarray, c: Context) -> np.ndarray:
    """shims.py:118-268."""
    ubj = c.front + rotate(c.front_to_ubj, c.axis * x[0])
    rc, ru = x[1:4], x[4:7]
    n_c, n_u = rotate(c.n0, rc), rotate(c.n0, ru)
    ca, cb = ubj + rotate(c.ua, rc), ubj + rotate(c.ub, rc)
    ba, bb = c.lbj + rotate(c.la, ru), c.lbj + rotate(c.lb, ru)
    out = [ba - ca - c.t * n_c, bb - cb - c.t * n_c, n_u - n_c,
           np.array([np.linalg.norm(c.lbj + rotate(c.l_hl, ru) - c.hl_in) - c.hl_len])]
    if c.rocker is not None:
        a, direction, to_pi, to_po, length = c.rocker
        pi = a + rotate(to_pi, direction * x[7])
        po = c.lbj + rotate(to_po, ru)
        out.append(np.array([np.linalg.norm(po - pi) - length]))
    return np.concatenate(out)


@dataclass
class Solution:
    x: np.ndarray
    ubj: np.ndarray
    upright_rotvec: np.ndarray
    rocker_angle: float
    residual_norm: float
    max_residual: float
    success: bool


def solve(c: Context, tight: bool = False) -> Solution:
    """shims.py:442-470; ``tight`` drives MINPACK to machine precision (what the device iterates to)."""
    n = 8 if c.rocker is not None else 7
    tol = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15) if tight else dict(ftol=1e-5, xtol=1e-9, gtol=1e-9)
    result = least_squares(residuals, np.zeros(n), args=(c,), method="lm", **tol)
    x = result.x
    return Solution(
        x=x, ubj=c.front + rotate(c.front_to_ubj, c.axis * x[0]), upright_rotvec=x[4:7].copy(),
        rocker_angle=float(x[7]) if n == 8 else 0.0, residual_norm=float(np.linalg.norm(result.fun)),
        max_residual=float(np.max(np.abs(result.fun))), success=bool(result.success))


def apply(points: dict, shim: ShimConfig, roles: dict, tight: bool = False) -> tuple[dict, Solution | None]:
    """
    ``DoubleWishboneSuspension.apply_camber_shim`` (double_wishbone.py:501-570).  ``roles``: ubj, lbj,
    uwb_front, uwb_rear, heading_inboard, heading_outboard (point names), upright_points (names rotated about
    the LBJ), rocker (None or dict axis_a / axis_b / pushrod_inboard / pushrod_outboard) and rocker_points.
    Returns the setup positions (a new dict) and the assembly solution (None on the equal-thickness exit).
    """
    out = {k: np.asarray(v, float).copy() for k, v in points.items()}
    if abs(shim.setup_thickness - shim.design_thickness) < EPS_GEOMETRIC:  # shims.py:346-357
        return out, None
    pos = {"ubj": out[roles["ubj"]], "lbj": out[roles["lbj"]], "uwb_front": out[roles["uwb_front"]],
           "uwb_rear": out[roles["uwb_rear"]], **out}
    c = build_context(pos, shim, roles["heading_inboard"], roles["heading_outboard"], roles.get("rocker"))
    sol = solve(c, tight)
    out[roles["ubj"]] = sol.ubj
    angle = float(np.linalg.norm(sol.upright_rotvec))
    if angle > EPS_GEOMETRIC:  # double_wishbone.py:551-562
        axis = sol.upright_rotvec / angle
        for name in roles["upright_points"]:
            out[name] = rotate_about_axis(out[name], out[roles["lbj"]], axis, angle)
    if roles.get("rocker") is not None:  # mechanisms.py:247-265
        a, b = out[roles["rocker"]["axis_a"]], out[roles["rocker"]["axis_b"]]
        axis = (b - a) / float(np.linalg.norm(b - a))
        for name in dict.fromkeys(roles["rocker_points"]):
            out[name] = rotate_about_axis(out[name], a, axis, sol.rocker_angle)
    return out, sol
