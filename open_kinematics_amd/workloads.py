"""
The BASELINE.json configurations as synthetic inputs.  Geometry comes from
the YAML fixtures the reference's own tests hold, shipped as package data under
``open_kinematics_amd/data`` (the same files the parity tests keep under ``tests/golden/geometry``).
"""

from __future__ import annotations

import os

import numpy as np

from .enums import Axis, PointID, Side
from .input import build_suspension, load_geometry
from .sweep import target_rows
from .targeting import PointTargetAxis

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
P = PointID
Y, Z = PointTargetAxis(Axis.Y), PointTargetAxis(Axis.Z)


def geometry_path(name: str) -> str:
    return os.path.join(_DATA, name)


def bump_sweep_problem(n_steps: int, line_mode: str = "pinned"):
    """C1 / C2: double-wishbone corner, rack held, wheel centre Z swept -60..+80 mm."""
    sus = load_geometry(geometry_path("geometry.yaml"))
    program, base = target_rows(sus, [(P.TRACKROD_INBOARD, Y), (P.WHEEL_CENTER, Z)], line_mode)
    targets = np.stack([np.full(n_steps, base[0]), base[1] + np.linspace(-60.0, 80.0, n_steps)], axis=1)
    return program, np.ascontiguousarray(targets)


def axle_grid_problem(n_heave: int = 256, n_roll: int = 256, line_mode: str = "pinned"):
    """C3: rocker + U-bar axle, heave (+-30) x roll (+-20) grid flattened row-major."""
    sus = load_geometry(geometry_path("axle_geometry_rocker.yaml"))
    L, R = Side.LEFT, Side.RIGHT
    from .enums import PointRef

    specs = [(PointRef(L, P.WHEEL_CENTER), Z), (PointRef(R, P.WHEEL_CENTER), Z), (PointRef(L, P.TRACKROD_INBOARD), Y)]
    program, base = target_rows(sus, specs, line_mode)
    h, r = np.meshgrid(np.linspace(-30.0, 30.0, n_heave), np.linspace(-20.0, 20.0, n_roll), indexing="ij")
    targets = np.stack([base[0] + (h + r).ravel(), base[1] + (h - r).ravel(), np.full(h.size, base[2])], axis=1)
    return program, np.ascontiguousarray(targets)


def macpherson_grid_problem(n_bump: int = 512, n_rack: int = 512, line_mode: str = "pinned"):
    """C4: MacPherson corner, bump (-60..80) x rack (+-40) grid flattened row-major."""
    sus = load_geometry(geometry_path("macpherson_geometry.yaml"))
    program, base = target_rows(sus, [(P.TRACKROD_INBOARD, Y), (P.WHEEL_CENTER, Z)], line_mode)
    b, k = np.meshgrid(np.linspace(-60.0, 80.0, n_bump), np.linspace(-40.0, 40.0, n_rack), indexing="ij")
    targets = np.stack([base[0] + k.ravel(), base[1] + b.ravel()], axis=1)
    return program, np.ascontiguousarray(targets)


def ensemble_problem(n_geometries: int = 4096, n_steps: int = 256, sigma: float = 1.0, seed: int = 0,
                     line_mode: str = "pinned"):
    """
    C5: perturbed double-wishbone hardpoints (every authored coordinate + N(0, sigma)) x bump
    sweep.  Returns the base program, hardpoint table ``[G, P, 3]`` (authored points perturbed,
    derived entries recomputed on device by ``rebind``) and relative targets ``[S, T]``.
    Draws that violate the loader's validators are redrawn: the side sign
    (``suspensions/build.py:322-343``: a left corner needs ``AXLE_OUTBOARD`` y > 0) and the track rod's rigid
    anchors (``corner/attachments.py:77-94`` through ``corner/track_rod.py:51-53``: the first two upright
    anchors distinct, the third off their line, both to ``EPS_GEOMETRIC``).
    """
    import yaml

    from .topology import EPS_GEOMETRIC

    with open(geometry_path("geometry.yaml"), "r", encoding="utf-8") as fh:
        base_map = yaml.safe_load(fh)
    sus = build_suspension(base_map)
    program, _ = target_rows(sus, [(P.TRACKROD_INBOARD, Y), (P.WHEEL_CENTER, Z)], line_mode)
    authored = [program.point_index(k) for k in sus.hardpoints]
    axle_outboard = program.point_index(P.AXLE_OUTBOARD)
    anchors = [program.point_index(k) for k in sus.UPRIGHT_BODY[:3]]

    def valid(points: np.ndarray) -> bool:
        if points[axle_outboard, 1] <= 0.0:
            return False
        a, b, c = points[anchors]
        span = float(np.linalg.norm(b - a))
        if span <= EPS_GEOMETRIC:
            return False
        return float(np.linalg.norm(np.cross(c - a, (b - a) / span))) > EPS_GEOMETRIC

    rng = np.random.default_rng(seed)
    table = np.repeat(program.design_pos[None], n_geometries, axis=0)
    g = 0
    while g < n_geometries:
        trial = program.design_pos.copy()
        trial[authored] += rng.normal(0.0, sigma, (len(authored), 3))
        if not valid(trial):
            continue
        table[g] = trial
        g += 1
    relative = np.stack([np.zeros(n_steps), np.linspace(-60.0, 80.0, n_steps)], axis=1)
    return program, table, relative
