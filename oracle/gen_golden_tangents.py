"""
Golden fixtures for the solution-manifold tangents by RUNNING the real reference:
``kinematics.core.sensitivity.compute_state_tangents`` at tight-tolerance solved states.

Run here (where /root/reference exists):  python -m oracle.gen_golden_tangents
Only data is written: tests/golden/tangents_<name>.npz with, for K sampled sweep steps,
  step_index [K], pos [K, n_out, 3] (the reference's tight solved positions),
  vel [K, T, n_out, 3] (d point / d target), rank [K], smallest_sv [K], cond [K].
Inputs (programs, targets) are the ones of tests/golden/<name>.npz.
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
from kinematics.core.points.derived.manager import DerivedPointsManager  # noqa: E402
from kinematics.core.sensitivity import compute_state_tangents  # noqa: E402
from kinematics.core.solver import SolverConfig, convert_targets_to_absolute, solve_suspension_sweep  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
TIGHT = SolverConfig(ftol=1e-15, xtol=1e-15, gtol=1e-15)


def emit(name: str, n_samples: int) -> None:
    base = np.load(os.path.join(OUT, f"{name}.npz"), allow_pickle=False)
    geometry = yaml.safe_load(str(base["geometry_yaml"]))
    sweep_map = yaml.safe_load(str(base["sweep_yaml"]))
    suspension = build_suspension(copy.deepcopy(geometry))
    sweep = build_sweep(sweep_map, suspension)
    manager = DerivedPointsManager(suspension.derived_spec())
    states, _ = solve_suspension_sweep(suspension.initial_state(), suspension.constraints(), sweep, manager, TIGHT)
    out = suspension.output_points()
    initial = suspension.initial_state()
    idx = np.unique(np.linspace(0, sweep.n_steps - 1, n_samples).round().astype(int))
    pos, vel, rank, ssv, cond = [], [], [], [], []
    for i in idx:
        step_targets = convert_targets_to_absolute([s[i] for s in sweep.target_sweeps], initial)
        fields, info = compute_state_tangents(states[i], suspension.constraints(), manager, step_targets)
        pos.append([states[i].positions[k].data for k in out])
        vel.append([[f.velocity(k) for k in out] for f in fields])
        rank.append(info.rank)
        ssv.append(info.smallest_singular_value)
        cond.append(info.condition_number)
    np.savez_compressed(
        os.path.join(OUT, f"tangents_{name}.npz"),
        step_index=idx, pos=np.asarray(pos, dtype=np.float64), vel=np.asarray(vel, dtype=np.float64),
        rank=np.asarray(rank, dtype=np.int32), smallest_sv=np.asarray(ssv), cond=np.asarray(cond),
    )
    print(f"tangents_{name}: {len(idx)} states, T={len(sweep.target_sweeps)}, "
          f"cond max {max(cond):.3g}, |vel| max {np.abs(vel).max():.3g}")


def main() -> None:
    emit("c1_dw_corner", 21)
    emit("c4_macpherson_grid", 24)
    emit("u_dw_corner", 8)
    emit("u_macpherson", 8)
    emit("c3_axle_grid", 6)


if __name__ == "__main__":
    main()
