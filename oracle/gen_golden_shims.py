"""
Golden fixtures for the camber-shim setup solve by RUNNING the real reference:
``DoubleWishboneSuspension.initial_state()`` with a ``camber_shim`` whose setup thickness differs from
its design thickness (-> ``apply_camber_shim`` -> ``solve_camber_shim_assembly``).

Run here (where /root/reference exists):  python -m oracle.gen_golden_shims
Writes tests/golden/shims_<case>.npz: geometry_yaml (the reference's test geometry, shim block
included), names [P] of the state's points (authored + derived), authored [P, 3] (design state with
setup == design), setup [K] thicknesses, positions [K, P, 3] (the reference's setup states), and the
assembly solution per thickness (ubj, upright_rotvec, rocker_angle, residual_norm).
"""

from __future__ import annotations

import copy
import os
import sys

import numpy as np
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import ref_shim  # noqa: E402

ref_shim.install()

from kinematics.core.enums import PointID  # noqa: E402
from kinematics.core.input import build_suspension  # noqa: E402
from kinematics.core.suspensions.config.shims import (  # noqa: E402
    CamberShimRockerCoupling,
    solve_camber_shim_assembly,
)

OUT = os.path.join(REPO, "tests", "golden")
REF_DATA = "/root/reference/tests/data"
SHIM = {
    "shim_face_point_a": {"x": -25.0, "y": 750.0, "z": 510.0},
    "shim_face_point_b": {"x": -25.0, "y": 750.0, "z": 490.0},
    "shim_face_normal": {"x": 0.0, "y": 1.0, "z": 0.0},
    "design_thickness": 30.0,
}


def emit(case: str, geometry_file: str, setups: list[float]) -> None:
    with open(os.path.join(REF_DATA, geometry_file), encoding="utf-8") as fh:
        geometry = yaml.safe_load(fh)
    geometry["config"]["camber_shim"] = {**SHIM, "setup_thickness": SHIM["design_thickness"]}
    design = build_suspension(copy.deepcopy(geometry))
    state = design.initial_state()
    keys = list(state.positions)
    names = [k.name.lower() for k in keys]
    authored = np.asarray([state.positions[k].data for k in keys])
    positions, ubj, rotvec, rocker, resid = [], [], [], [], []
    for t in setups:
        g = copy.deepcopy(geometry)
        g["config"]["camber_shim"]["setup_thickness"] = float(t)
        sus = build_suspension(g)
        st = sus.initial_state()
        positions.append([st.positions[k].data for k in keys])
        upright_pushrod = "pushrod_outboard" in names and str(geometry.get("actuation", {}).get("mount")) == "upright"
        coupling = CamberShimRockerCoupling(PointID.ROCKER_AXIS_A, PointID.ROCKER_AXIS_B, PointID.PUSHROD_INBOARD,
                                            PointID.PUSHROD_OUTBOARD) if upright_pushrod else None
        sol = solve_camber_shim_assembly(sus.get_hardpoints_copy(), sus.config.camber_shim,
                                         sus.wheel_heading_link.inboard_point, sus.wheel_heading_link.outboard_point,
                                         rocker_coupling=coupling)
        ubj.append(sol.ubj_position)
        rotvec.append(sol.upright_body_rot_vec)
        rocker.append(sol.rocker_angle_rad)
        resid.append(sol.constraint_residual_norm)
    np.savez_compressed(
        os.path.join(OUT, f"shims_{case}.npz"),
        geometry_yaml=yaml.safe_dump(geometry, sort_keys=False), names=np.array(names), authored=authored,
        setup=np.asarray(setups, dtype=np.float64), positions=np.asarray(positions), ubj=np.asarray(ubj),
        upright_rotvec=np.asarray(rotvec), rocker_angle=np.asarray(rocker), residual_norm=np.asarray(resid),
        upright_points=np.array([p.name.lower() for p in design.upright_attachment_points()]),
    )
    moved = np.max(np.abs(np.asarray(positions) - authored[None]), axis=(0, 2))
    print(f"shims_{case}: {len(setups)} thicknesses, {len(names)} points; moved:",
          {n: round(float(m), 3) for n, m in zip(names, moved) if m > 1e-9}, "residual norms", np.round(resid, 12))


def emit_axle(case: str, geometry_file: str, setups: list[float]) -> None:
    """Axle: the shim is authored on the left setup; the right corner gets the mirrored shim (build.py:310-318,357-375)."""
    from kinematics.core.primitives.point_ref import PointRef

    with open(os.path.join(REF_DATA, geometry_file), encoding="utf-8") as fh:
        geometry = yaml.safe_load(fh)
    geometry["axle_config"]["left_setup"] = {"camber_shim": {**SHIM, "setup_thickness": SHIM["design_thickness"]}}
    design = build_suspension(copy.deepcopy(geometry)).initial_state()
    keys = list(design.positions)
    names = [f"{k.side.name.lower()}_{k.point.name.lower()}" if isinstance(k, PointRef) else k.name.lower() for k in keys]
    positions = []
    for t in setups:
        g = copy.deepcopy(geometry)
        g["axle_config"]["left_setup"]["camber_shim"]["setup_thickness"] = float(t)
        st = build_suspension(g).initial_state()
        positions.append([st.positions[k].data for k in keys])
    authored = np.asarray([design.positions[k].data for k in keys])
    np.savez_compressed(os.path.join(OUT, f"shims_{case}.npz"), geometry_yaml=yaml.safe_dump(geometry, sort_keys=False),
                        names=np.array(names), authored=authored, setup=np.asarray(setups, dtype=np.float64),
                        positions=np.asarray(positions))
    moved = np.max(np.abs(np.asarray(positions) - authored[None]), axis=(0, 2))
    print(f"shims_{case}: {len(setups)} thicknesses, {len(names)} points; moved:",
          {n: round(float(m), 3) for n, m in zip(names, moved) if m > 1e-9})


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "axle":
        emit_axle("axle_rocker", "axle_geometry_rocker.yaml", [26.0, 30.0, 38.0])
        return
    emit("dw", "geometry.yaml", [20.0, 25.0, 29.0, 30.0, 31.0, 35.0, 40.0, 45.0])
    emit("dw_rocker", "corner_strut_rocker_geometry.yaml", [22.0, 30.0, 36.0, 40.0])
    emit_axle("axle_rocker", "axle_geometry_rocker.yaml", [26.0, 30.0, 38.0])


if __name__ == "__main__":
    main()
