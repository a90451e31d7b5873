"""
Golden fixture covering ALL 13 constraint classes of the reference plus target rows
(five of them are used by no BASELINE topology) by RUNNING the real reference:
a synthetic 3-fixed / 5-free point problem with one row of every class,
``ResidualComputer.compute`` / ``compute_jacobian`` at seeded free vectors.

Run here (where /root/reference exists):  python -m oracle.gen_golden_rows
Writes tests/golden/rows_all_classes.npz in the layout of the other fixtures (prog_*, eval_*).
"""

from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import ref_shim  # noqa: E402

ref_shim.install()

from kinematics.core import constraints as rc  # noqa: E402
from kinematics.core.enums import Axis, PointID, TargetPositionMode  # noqa: E402
from kinematics.core.points.derived.manager import DerivedPointsManager, DerivedPointsSpec  # noqa: E402
from kinematics.core.primitives.geometry import Direction3, Point3  # noqa: E402
from kinematics.core.solver import ResidualComputer  # noqa: E402
from kinematics.core.state import SuspensionState  # noqa: E402
from kinematics.core.targeting import PointTarget, PointTargetAxis  # noqa: E402

from open_kinematics_amd.program import flatten_problem  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
P = PointID


def main() -> None:
    rng = np.random.default_rng(7)
    fixed = [P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.UPPER_WISHBONE_INBOARD_FRONT]
    free = [P.LOWER_WISHBONE_OUTBOARD, P.UPPER_WISHBONE_OUTBOARD, P.TRACKROD_OUTBOARD, P.AXLE_INBOARD, P.AXLE_OUTBOARD]
    positions = {k: Point3(rng.uniform(-300.0, 300.0, 3)) for k in fixed + free}
    state = SuspensionState(positions=positions, free_points=set(free))
    a, b, c = fixed
    f0, f1, f2, f3, f4 = free
    unit = lambda v: Direction3(np.asarray(v, dtype=float) / np.linalg.norm(v))  # noqa: E731
    constraints = [
        rc.DistanceConstraint(a, f0, 410.0),
        rc.DistanceConstraint(f0, f1, 333.0),
        rc.SphericalJointConstraint(f2, f3),
        rc.AngleConstraint(a, f0, f1, f4, 0.9),
        rc.ThreePointAngleConstraint(f0, f1, b, 1.3),
        rc.VectorsParallelConstraint(a, f1, f2, f4),
        rc.VectorsPerpendicularConstraint(f0, f3, b, f4),
        rc.EqualDistanceConstraint(a, f2, f3, f4),
        rc.FixedAxisConstraint(f3, Axis.Y, 12.5),
        rc.PointOnLineConstraint(f2, Point3([10.0, 20.0, 30.0]), unit([1.0, 2.0, -0.5])),
        rc.PointOnPlaneConstraint(f4, Point3([-5.0, 15.0, 40.0]), unit([0.3, -1.0, 0.2])),
        rc.MidpointOnPlaneConstraint(f0, f4, Point3([0.0, 0.0, 50.0]), unit([0.0, 0.2, 1.0])),
        rc.CoplanarPointsConstraint(a, f0, f1, f2),
        rc.ScalarTripleProductConstraint(c, f1, f3, f4, 1.5e6),
        rc.DistanceConstraint(b, f3, 280.0),
        rc.DistanceConstraint(c, f4, 390.0),
    ]
    targets = [
        PointTarget(point_id=f1, direction=PointTargetAxis(axis=Axis.Z), value=35.0, mode=TargetPositionMode.ABSOLUTE),
        PointTarget(point_id=f4, direction=PointTargetAxis(axis=Axis.X), value=-40.0, mode=TargetPositionMode.ABSOLUTE),
    ]
    spec = DerivedPointsSpec(functions={}, dependencies={})
    computer = ResidualComputer(constraints, DerivedPointsManager(spec), state.copy(), len(targets))
    x0 = state.get_free_array()
    xs, rs, js = [], [], []
    for k in range(12):
        x = x0 + (rng.normal(0.0, 5.0, x0.shape) if k > 0 else 0.0)
        xs.append(x.copy())
        rs.append(computer.compute(x.copy(), targets))
        js.append(computer.compute_jacobian(x.copy(), targets))
    program = flatten_problem(state, constraints, spec, [(t.point_id, t.direction) for t in targets],
                              list(positions.keys()), line_mode="softnorm")
    arrays = {f"prog_{k}": v for k, v in program.to_arrays().items()}
    arrays.update(eval_x=np.asarray(xs), eval_targets=np.tile([[t.value for t in targets]], (12, 1)),
                  eval_r=np.asarray(rs), eval_jac=np.asarray(js))
    np.savez_compressed(os.path.join(OUT, "rows_all_classes.npz"), **arrays)
    kinds = sorted({int(t) for t in program.row_type})
    print(f"rows_all_classes: n={program.n_vars} m={program.n_residuals}, row types {kinds}")


if __name__ == "__main__":
    main()
