"""
Golden fixtures for the state metrics and derivative columns by RUNNING the
real reference: ``kinematics.core.sweep.compute_sweep_metrics`` on its own default-tolerance states.

Run here (where /root/reference exists):  python -m oracle.gen_golden_metrics
Writes tests/golden/metrics_<name>.npz: pos [S, n_out, 3] (the states the metrics belong to),
values [S, 19] in OKX_METRIC_* order (None -> NaN), deriv_names, deriv [S, n_deriv], side_sign, role
point names, the instant-axis construction, damper points and the vehicle numbers the anti-geometry
reads.  ``*_anti`` variants author front_brake_bias / axle_position / driven_axle so that anti-dive,
anti-lift and anti-squat are defined; ``metrics_axle_c3`` holds the axle-scope row and both corner rows.
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
from kinematics.core.input import build_suspension, build_sweep  # noqa: E402
from kinematics.core.primitives.point_ref import Side  # noqa: E402
from kinematics.core.sweep import compute_sweep_metrics, solve_sweep  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
NAMES = ("camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius", "mechanical_trail",
         "svic_x", "svic_z", "svsa_length", "fvic_y", "fvic_z", "fvsa_length", "damper_length", "svsa_angle",
         "anti_dive", "anti_lift", "anti_squat")
AXLE_NAMES = ("heave", "roll", "ride_height_change", "track", "roll_center_y", "roll_center_z", "rack_displacement")
DW_AXIS = ("upper_wishbone_inboard_front", "upper_wishbone_inboard_rear", "upper_wishbone_outboard",
           "lower_wishbone_inboard_front", "lower_wishbone_inboard_rear", "lower_wishbone_outboard")
MAC_AXIS = ("lower_wishbone_inboard_front", "lower_wishbone_inboard_rear", "lower_wishbone_outboard", "strut_top")


def _num(v) -> float:
    return float("nan") if v is None else float(v)


def _corner_meta(corner, config) -> dict:
    axle_in, axle_out = corner.wheel_axis_points()
    lower, upper = corner.steering_axis_points()
    kind = type(corner).__name__.lower()
    damper = corner.damper_points()
    return dict(
        side_sign=float(corner.side.lateral_sign),
        roles=np.array([axle_in.name, axle_out.name, lower.name, upper.name]),
        axis_kind="two_planes" if "wishbone" in kind else "plane_and_strut",
        axis_points=np.array(DW_AXIS if "wishbone" in kind else MAC_AXIS),
        damper=np.array([] if damper is None else [damper[0].name.lower(), damper[1].name.lower()]),
        wheelbase=float(config.wheelbase), cg_z=float(config.cg_position[2]),
        front_brake_bias=_num(config.front_brake_bias),
        axle_position="" if config.axle_position is None else str(config.axle_position.value),
        driven_axle="" if config.driven_axle is None else str(config.driven_axle.value),
    )


def emit(name: str, out_name: str | None = None, config_patch: dict | None = None, stride: int = 1) -> None:
    base = np.load(os.path.join(OUT, f"{name}.npz"), allow_pickle=False)
    geometry = yaml.safe_load(str(base["geometry_yaml"]))
    if config_patch:
        geometry["config"].update(config_patch)
    sweep_map = yaml.safe_load(str(base["sweep_yaml"]))
    suspension = build_suspension(copy.deepcopy(geometry))
    sweep = build_sweep(sweep_map, suspension)
    states, _ = solve_sweep(suspension, sweep)
    result = compute_sweep_metrics(suspension, sweep, states)
    assert result.derivative_error is None, result.derivative_error
    out = suspension.output_points()
    keep = range(0, len(states), stride)
    pos = np.asarray([[states[s].positions[k].data for k in out] for s in keep], dtype=np.float64)
    rows = [result.rows[s] for s in keep]
    values = np.asarray([[_num(row[n]) for n in NAMES] for row in rows], dtype=np.float64)
    deriv_names = [k for k in rows[0] if k.startswith("deriv_")]
    deriv = np.asarray([[row[k] for k in deriv_names] for row in rows], dtype=np.float64)
    np.savez_compressed(
        os.path.join(OUT, f"metrics_{out_name or name}.npz"),
        pos=pos, values=values, deriv=deriv, deriv_names=np.array(deriv_names),
        **_corner_meta(suspension, suspension.config),
    )
    defined = [n for k, n in enumerate(NAMES) if np.isfinite(values[:, k]).all()]
    print(f"metrics_{out_name or name}: {len(rows)} states, defined: {defined}; derivative columns: {deriv_names}")


def emit_axle(name: str, out_name: str, stride: int) -> None:
    base = np.load(os.path.join(OUT, f"{name}.npz"), allow_pickle=False)
    geometry = yaml.safe_load(str(base["geometry_yaml"]))
    sweep_map = yaml.safe_load(str(base["sweep_yaml"]))
    axle = build_suspension(copy.deepcopy(geometry))
    sweep = build_sweep(sweep_map, axle)
    states, _ = solve_sweep(axle, sweep)
    result = compute_sweep_metrics(axle, sweep, states)
    out = axle.output_points()
    keep = range(0, len(states), stride)
    pos = np.asarray([[states[s].positions[k].data for k in out] for s in keep], dtype=np.float64)
    rows = [result.rows[s] for s in keep]
    arrays = dict(pos=pos, axle_values=np.asarray([[_num(r.axle[n]) for n in AXLE_NAMES] for r in rows]))
    for side in (Side.LEFT, Side.RIGHT):
        tag = side.name.lower()
        corner = axle.corners[side]
        arrays[f"{tag}_values"] = np.asarray([[_num(r.corners[side][n]) for n in NAMES] for r in rows])
        meta = _corner_meta(corner, corner.config if corner.config is not None else axle.config)
        arrays.update({f"{tag}_{k}": v for k, v in meta.items()})
        rack = corner.rack_attachment_point()
        arrays[f"{tag}_rack"] = "" if rack is None else rack.name.lower()
        # topology-specific extras (rocker_angle, torsion_bar_twist, arb_arm_angle) and every derivative column
        extra = [k for k in rows[0].corners[side] if k not in NAMES and not k.startswith("deriv_")]
        arrays[f"{tag}_extra_names"] = np.array(extra)
        arrays[f"{tag}_extra_values"] = np.asarray([[_num(r.corners[side][k]) for k in extra] for r in rows])
        dnames = [k for k in rows[0].corners[side] if k.startswith("deriv_")]
        arrays[f"{tag}_deriv_names"] = np.array(dnames)
        arrays[f"{tag}_deriv"] = np.asarray([[_num(r.corners[side][k]) for k in dnames] for r in rows])
    extra = [k for k in rows[0].axle if k not in AXLE_NAMES and not k.startswith("deriv_")]
    arrays["axle_extra_names"] = np.array(extra)
    arrays["axle_extra_values"] = np.asarray([[_num(r.axle[k]) for k in extra] for r in rows])
    dnames = [k for k in rows[0].axle if k.startswith("deriv_")]
    arrays["axle_deriv_names"] = np.array(dnames)
    arrays["axle_deriv"] = np.asarray([[_num(r.axle[k]) for k in dnames] for r in rows])
    np.savez_compressed(os.path.join(OUT, f"metrics_{out_name}.npz"), **arrays)
    arrays["axle_key_order"] = np.array(list(rows[0].axle))
    arrays["left_key_order"] = np.array(list(rows[0].corners[Side.LEFT]))
    print(f"   extras: axle {extra} + {dnames}; left {list(arrays['left_extra_names'])} + {list(arrays['left_deriv_names'])}")
    print(f"metrics_{out_name}: {len(rows)} states; axle row {dict(zip(AXLE_NAMES, arrays['axle_values'][len(rows) // 3]))}")


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "axle":
        emit_axle("c3_axle_grid", "axle_c3", stride=5)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "hardware":  # T-bar anti-roll bar, rocker-to-rocker heave link
        emit_axle("t_axle_t_bar_roll", "axle_t_bar_roll", stride=3)
        emit_axle("t_axle_t_bar_bump", "axle_t_bar_bump", stride=5)
        emit_axle("t_axle_heave_link", "axle_heave_link", stride=1)
        return
    emit("c1_dw_corner")
    emit("c4_macpherson_grid")
    emit("e2e_sweep")
    emit("c1_dw_corner", "dw_front_anti", {"front_brake_bias": 0.65, "axle_position": "front", "driven_axle": "front"}, stride=4)
    emit("c4_macpherson_grid", "mac_rear_anti", {"front_brake_bias": 0.6, "axle_position": "rear", "driven_axle": "rear"}, stride=5)
    emit_axle("c3_axle_grid", "axle_c3", stride=5)
    _ = PointID


if __name__ == "__main__":
    main()
