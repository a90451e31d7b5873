"""
Golden fixtures for the corner state metrics and derivative columns by RUNNING the
real reference: ``kinematics.core.sweep.compute_sweep_metrics`` on its own default-tolerance states.

Run here (where /root/reference exists):  python -m oracle.gen_golden_metrics
Writes tests/golden/metrics_<name>.npz: pos [S, n_out, 3] (the states the metrics belong to),
values [S, 8] in OKX_METRIC_* order, deriv_names, deriv [S, n_deriv], side_sign, role point names.
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

from kinematics.core.input import build_suspension, build_sweep  # noqa: E402
from kinematics.core.sweep import compute_sweep_metrics, solve_sweep  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
NAMES = ("camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius", "mechanical_trail")


def emit(name: str) -> None:
    base = np.load(os.path.join(OUT, f"{name}.npz"), allow_pickle=False)
    geometry = yaml.safe_load(str(base["geometry_yaml"]))
    sweep_map = yaml.safe_load(str(base["sweep_yaml"]))
    suspension = build_suspension(copy.deepcopy(geometry))
    sweep = build_sweep(sweep_map, suspension)
    states, _ = solve_sweep(suspension, sweep)
    result = compute_sweep_metrics(suspension, sweep, states)
    assert result.derivative_error is None, result.derivative_error
    out = suspension.output_points()
    pos = np.asarray([[s.positions[k].data for k in out] for s in states], dtype=np.float64)
    values = np.asarray([[row[n] for n in NAMES] for row in result.rows], dtype=np.float64)
    deriv_names = [k for k in result.rows[0] if k.startswith("deriv_")]
    deriv = np.asarray([[row[k] for k in deriv_names] for row in result.rows], dtype=np.float64)
    axle_in, axle_out = suspension.wheel_axis_points()
    lower, upper = suspension.steering_axis_points()
    np.savez_compressed(
        os.path.join(OUT, f"metrics_{name}.npz"),
        pos=pos, values=values, deriv=deriv, deriv_names=np.array(deriv_names),
        side_sign=float(suspension.side.lateral_sign),
        roles=np.array([axle_in.name, axle_out.name, lower.name, upper.name]),
    )
    print(f"metrics_{name}: {len(states)} states, derivative columns: {deriv_names}")


def main() -> None:
    emit("c1_dw_corner")
    emit("c4_macpherson_grid")
    emit("e2e_sweep")


if __name__ == "__main__":
    main()
