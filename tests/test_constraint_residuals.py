"""
CPU: the host-side ``Constraint.residual(positions)`` of all 13 descriptor classes (reference ``core/constraints.py:56-733``)
against ``tests/golden/rows_all_classes.npz`` - rows and residuals generated from the reference's real
``ResidualComputer`` (``oracle/gen_golden_rows.py``): one row of every class, twelve seeded states.
"""

import numpy as np

from conftest import load_golden


def _constraint_of_row(row_type: int, pts, q, keys):
    from open_kinematics_amd import constraints as C
    from open_kinematics_amd import program as P

    k = [keys[i] for i in pts if i >= 0]
    make = {
        P.ROW_DISTANCE: lambda: C.DistanceConstraint(k[0], k[1], q[0]),
        P.ROW_SPHERICAL: lambda: C.SphericalJointConstraint(k[0], k[1]),
        P.ROW_ANGLE: lambda: C.AngleConstraint(k[0], k[1], k[2], k[3], q[0]),
        P.ROW_THREE_POINT_ANGLE: lambda: C.ThreePointAngleConstraint(k[0], k[1], k[2], q[0]),
        P.ROW_VECTORS_PARALLEL: lambda: C.VectorsParallelConstraint(*k[:4]),
        P.ROW_VECTORS_PERPENDICULAR: lambda: C.VectorsPerpendicularConstraint(*k[:4]),
        P.ROW_EQUAL_DISTANCE: lambda: C.EqualDistanceConstraint(*k[:4]),
        P.ROW_FIXED_AXIS: lambda: C.FixedAxisConstraint(k[0], int(q[0]), q[1]),
        P.ROW_POINT_ON_LINE: lambda: C.PointOnLineConstraint(k[0], q[0:3], q[3:6]),
        P.ROW_POINT_ON_PLANE: lambda: C.PointOnPlaneConstraint(k[0], q[0:3], q[3:6]),
        P.ROW_MIDPOINT_ON_PLANE: lambda: C.MidpointOnPlaneConstraint(k[0], k[1], q[0:3], q[3:6]),
        P.ROW_COPLANAR: lambda: C.CoplanarPointsConstraint(*k[:4]),
        P.ROW_SCALAR_TRIPLE: lambda: C.ScalarTripleProductConstraint(k[0], k[1], k[2], k[3], q[0], q[1]),
    }
    return make[int(row_type)]()


def test_every_constraint_class_reproduces_the_reference_residuals():
    from open_kinematics_amd.state import Point3

    arrays, program = load_golden("rows_all_classes")
    assert program.line_mode == "softnorm"  # the reference's literal rows
    keys = list(program.point_keys)
    n_c = program.n_rows
    constraints = [_constraint_of_row(program.row_type[i], program.row_pts[i], program.row_param[i], keys) for i in range(n_c)]
    assert len({type(c).__name__ for c in constraints}) == 13
    free = [int(p) for p in program.free_point]
    worst = 0.0
    for x, r in zip(arrays["eval_x"], arrays["eval_r"]):
        pos = np.array(program.design_pos, dtype=np.float64)
        pos[free] = x.reshape(-1, 3)
        as_arrays = {key: pos[i] for i, key in enumerate(keys)}
        as_points = {key: Point3(pos[i]) for i, key in enumerate(keys)}   # the reference's callers hand over Point3 objects
        for i, c in enumerate(constraints):
            got = c.residual(as_arrays)
            assert isinstance(got, float) and got == c.residual(as_points)
            worst = max(worst, abs(got - r[i]) / max(1.0, abs(r[i])))
    assert worst <= 1e-13, worst


def test_residual_survives_remap_and_the_base_class_is_abstract():
    import pytest

    from open_kinematics_amd.constraints import Constraint, DistanceConstraint, softnorm

    c = DistanceConstraint("a", "b", 5.0).remap(lambda key: ("left", key))
    state = {("left", "a"): np.zeros(3), ("left", "b"): np.array([3.0, 4.0, 0.0])}
    assert c.residual(state) == softnorm(25.0) - 5.0 and abs(c.residual(state)) < 2e-6   # (softnorm's -1e-6 offset)
    with pytest.raises(NotImplementedError):
        Constraint().residual(state)
