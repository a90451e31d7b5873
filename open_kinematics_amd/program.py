"""
Constraint program: the flat, array-based form of one suspension's solve problem.

The reference hands ``solve_suspension_sweep`` (reference ``core/solver.py:654-660``) a
``SuspensionState`` (dict of points), a ``list[Constraint]`` of Python objects, a
``DerivedPointsManager`` of Python callables and per-step ``PointTarget`` lists.  The
device solver needs the same information as plain arrays (``include/okx.h``,
``okx_program_desc``).  ``flatten_problem`` does that conversion by duck typing, so it
accepts the reference's own objects as well as this package's loader objects:

* constraint rows are recognised by class name and the attribute names the reference
  classes use (``core/constraints.py``),
* derived points are recognised by the name (and ``functools.partial`` keywords) of the
  function registered for them (``core/points/derived/definitions.py``),
* the variable order is ``sorted(free_points)`` exactly as ``core/state.py:46-50``.
"""

from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import Any, Iterable, Mapping, Sequence

import numpy as np

# ---- type codes: must match include/okx.h (tests/test_abi.py parses the header) ----
ROW_DISTANCE = 0
ROW_SPHERICAL = 1
ROW_ANGLE = 2
ROW_THREE_POINT_ANGLE = 3
ROW_VECTORS_PARALLEL = 4
ROW_VECTORS_PERPENDICULAR = 5
ROW_EQUAL_DISTANCE = 6
ROW_FIXED_AXIS = 7
ROW_POINT_ON_LINE = 8
ROW_POINT_ON_PLANE = 9
ROW_MIDPOINT_ON_PLANE = 10
ROW_COPLANAR = 11
ROW_SCALAR_TRIPLE = 12
ROW_LINE_PIN = 13

DOP_MIDPOINT = 0
DOP_ALONG = 1
DOP_CONTACT_PATCH = 2

ROW_PARAMS = 8
ROW_POINTS = 4
MAX_VARS = 126
MAX_ROWS = 128
MAX_POINTS = 96
MAX_TARGETS = 8

ROLE_FIXED = 0
ROLE_FREE = 1
ROLE_DERIVED = 2

ROW_TYPE_NAMES = {
    ROW_DISTANCE: "DistanceConstraint",
    ROW_SPHERICAL: "SphericalJointConstraint",
    ROW_ANGLE: "AngleConstraint",
    ROW_THREE_POINT_ANGLE: "ThreePointAngleConstraint",
    ROW_VECTORS_PARALLEL: "VectorsParallelConstraint",
    ROW_VECTORS_PERPENDICULAR: "VectorsPerpendicularConstraint",
    ROW_EQUAL_DISTANCE: "EqualDistanceConstraint",
    ROW_FIXED_AXIS: "FixedAxisConstraint",
    ROW_POINT_ON_LINE: "PointOnLineConstraint",
    ROW_POINT_ON_PLANE: "PointOnPlaneConstraint",
    ROW_MIDPOINT_ON_PLANE: "MidpointOnPlaneConstraint",
    ROW_COPLANAR: "CoplanarPointsConstraint",
    ROW_SCALAR_TRIPLE: "ScalarTripleProductConstraint",
    ROW_LINE_PIN: "LinePin",
}


def _raw(value: Any) -> np.ndarray:
    """Point3/Direction3-like (``.data``) or array-like -> float64[3]."""
    data = getattr(value, "data", value)
    arr = np.asarray(data, dtype=np.float64)
    if arr.shape != (3,):
        raise ValueError(f"expected a 3-vector, got shape {arr.shape}")
    return arr


def _base_name(key: Any) -> str:
    """Enum-style name of a point key without its side qualifier."""
    return getattr(key, "point", key).name


def key_name(key: Any) -> str:
    """Public lowercase point name (reference ``point_ref.py:point_key_name``)."""
    return str(key.name).lower()


@dataclass
class ConstraintProgram:
    """Array form of one geometry's solve problem (see ``include/okx.h``)."""

    point_keys: list[Any]
    role: np.ndarray  # int8 [P]
    free_point: np.ndarray  # int32 [F], variable-block order
    dop_type: np.ndarray  # int32 [D]
    dop_out: np.ndarray  # int32 [D]
    dop_pts: np.ndarray  # int32 [D,4]
    dop_param: np.ndarray  # f64 [D]
    row_type: np.ndarray  # int32 [Mc]
    row_pts: np.ndarray  # int32 [Mc,4]
    row_param: np.ndarray  # f64 [Mc,8]
    row_source: np.ndarray  # int32 [Mc] index into the original constraint list
    tgt_point: np.ndarray  # int32 [T]
    tgt_dir: np.ndarray  # f64 [T,3]
    out_point: np.ndarray  # int32 [n_out]
    design_pos: np.ndarray  # f64 [P,3]
    constraint_desc: list[str] = field(default_factory=list)
    target_desc: list[str] = field(default_factory=list)
    line_mode: str = "softnorm"

    # ---- sizes ----
    @property
    def n_points(self) -> int:
        return len(self.point_keys)

    @property
    def n_free(self) -> int:
        return int(self.free_point.shape[0])

    @property
    def n_vars(self) -> int:
        return 3 * self.n_free

    @property
    def n_derived(self) -> int:
        return int(self.dop_type.shape[0])

    @property
    def n_rows(self) -> int:
        return int(self.row_type.shape[0])

    @property
    def n_targets(self) -> int:
        return int(self.tgt_point.shape[0])

    @property
    def n_residuals(self) -> int:
        return self.n_rows + self.n_targets

    @property
    def n_out(self) -> int:
        return int(self.out_point.shape[0])

    def point_index(self, key: Any) -> int:
        return self._index()[key]

    def _index(self) -> dict[Any, int]:
        cache = getattr(self, "_index_cache", None)
        if cache is None:
            cache = {k: i for i, k in enumerate(self.point_keys)}
            object.__setattr__(self, "_index_cache", cache)
        return cache

    def design_free_array(self) -> np.ndarray:
        """x_0 of the reference (``state.py:59-72``)."""
        return self.design_pos[self.free_point].reshape(-1).copy()

    def with_targets(
        self, tgt_point: Sequence[int], tgt_dir: np.ndarray, target_desc: Sequence[str] = ()
    ) -> "ConstraintProgram":
        """Same problem with a different set of target rows."""
        clone = ConstraintProgram(
            point_keys=self.point_keys,
            role=self.role,
            free_point=self.free_point,
            dop_type=self.dop_type,
            dop_out=self.dop_out,
            dop_pts=self.dop_pts,
            dop_param=self.dop_param,
            row_type=self.row_type,
            row_pts=self.row_pts,
            row_param=self.row_param,
            row_source=self.row_source,
            tgt_point=np.asarray(tgt_point, dtype=np.int32).reshape(-1),
            tgt_dir=np.ascontiguousarray(np.asarray(tgt_dir, dtype=np.float64).reshape(-1, 3)),
            out_point=self.out_point,
            design_pos=self.design_pos,
            constraint_desc=self.constraint_desc,
            target_desc=list(target_desc),
            line_mode=self.line_mode,
        )
        return clone

    def with_line_mode(self, line_mode: str) -> "ConstraintProgram":
        """
        Re-express every point-on-line constraint in the requested form.

        ``"pinned"`` expands each ``ROW_POINT_ON_LINE`` row into three ``ROW_LINE_PIN``
        rows (components of ``(p - line_point) x line_dir``); ``"softnorm"`` collapses
        them back to the reference's scalar row (``constraints.py:560-576``).
        """
        if line_mode not in ("softnorm", "pinned"):
            raise ValueError("line_mode must be 'softnorm' or 'pinned'")
        rt, rp, rq, rs = [], [], [], []
        for i in range(self.n_rows):
            t = int(self.row_type[i])
            q = self.row_param[i].copy()
            if t == ROW_POINT_ON_LINE and line_mode == "pinned":
                for comp in range(3):
                    qq = q.copy()
                    qq[6] = float(comp)
                    rt.append(ROW_LINE_PIN)
                    rp.append(self.row_pts[i])
                    rq.append(qq)
                    rs.append(self.row_source[i])
            elif t == ROW_LINE_PIN and line_mode == "softnorm":
                if int(q[6]) == 0:
                    q[6] = 0.0
                    rt.append(ROW_POINT_ON_LINE)
                    rp.append(self.row_pts[i])
                    rq.append(q)
                    rs.append(self.row_source[i])
            else:
                rt.append(t)
                rp.append(self.row_pts[i])
                rq.append(q)
                rs.append(self.row_source[i])
        clone = self.with_targets(self.tgt_point, self.tgt_dir, self.target_desc)
        clone.row_type = np.asarray(rt, dtype=np.int32).reshape(-1)
        clone.row_pts = np.asarray(rp, dtype=np.int32).reshape(-1, ROW_POINTS)
        clone.row_param = np.asarray(rq, dtype=np.float64).reshape(-1, ROW_PARAMS)
        clone.row_source = np.asarray(rs, dtype=np.int32).reshape(-1)
        clone.line_mode = line_mode
        return clone

    def validate(self) -> None:
        """Static checks shared by the oracle and the device library."""
        if self.n_points > MAX_POINTS or self.n_vars > MAX_VARS:
            raise ValueError(
                f"problem too large: points={self.n_points} (limit {MAX_POINTS}), vars={self.n_vars} (limit {MAX_VARS}: "
                f"one thread per variable, two wavefronts per problem)"
            )
        if self.n_residuals > MAX_ROWS or self.n_targets > MAX_TARGETS:
            raise ValueError(
                f"problem too large: residual rows={self.n_residuals}, targets={self.n_targets}"
            )
        if self.n_vars > self.n_residuals:
            # reference: core/solver.py:116-121
            raise ValueError(
                f"System is underdetermined (n_vars={self.n_vars} > m_res={self.n_residuals}). "
                "The solve method (Levenberg-Marquardt) requires at least as "
                "many residuals as variables."
            )

    # ---- (de)serialisation for golden fixtures ----
    def to_arrays(self) -> dict[str, np.ndarray]:
        return {
            "point_names": np.array([key_name(k) for k in self.point_keys]),
            "role": self.role,
            "free_point": self.free_point,
            "dop_type": self.dop_type,
            "dop_out": self.dop_out,
            "dop_pts": self.dop_pts,
            "dop_param": self.dop_param,
            "row_type": self.row_type,
            "row_pts": self.row_pts,
            "row_param": self.row_param,
            "row_source": self.row_source,
            "tgt_point": self.tgt_point,
            "tgt_dir": self.tgt_dir,
            "out_point": self.out_point,
            "design_pos": self.design_pos,
            "constraint_desc": np.array(self.constraint_desc),
            "target_desc": np.array(self.target_desc),
            "line_mode": np.array(self.line_mode),
        }

    @classmethod
    def from_arrays(cls, arrays: Mapping[str, np.ndarray], prefix: str = "") -> "ConstraintProgram":
        def get(name: str) -> np.ndarray:
            return np.asarray(arrays[prefix + name])

        names = [str(s) for s in get("point_names")]
        return cls(
            point_keys=[NamedKey(s) for s in names],
            role=get("role").astype(np.int8),
            free_point=get("free_point").astype(np.int32),
            dop_type=get("dop_type").astype(np.int32),
            dop_out=get("dop_out").astype(np.int32),
            dop_pts=get("dop_pts").astype(np.int32).reshape(-1, 4),
            dop_param=get("dop_param").astype(np.float64),
            row_type=get("row_type").astype(np.int32),
            row_pts=get("row_pts").astype(np.int32).reshape(-1, 4),
            row_param=get("row_param").astype(np.float64).reshape(-1, ROW_PARAMS),
            row_source=get("row_source").astype(np.int32),
            tgt_point=get("tgt_point").astype(np.int32),
            tgt_dir=get("tgt_dir").astype(np.float64).reshape(-1, 3),
            out_point=get("out_point").astype(np.int32),
            design_pos=get("design_pos").astype(np.float64).reshape(-1, 3),
            constraint_desc=[str(s) for s in get("constraint_desc")],
            target_desc=[str(s) for s in get("target_desc")],
            line_mode=str(get("line_mode")),
        )


@dataclass(frozen=True, order=True)
class NamedKey:
    """Opaque point key used when a program is rebuilt from a fixture."""

    lower_name: str

    @property
    def name(self) -> str:
        return self.lower_name.upper()


# --------------------------------------------------------------------------------------
# Derived points
# --------------------------------------------------------------------------------------


def _unwrap_derived(function: Any) -> tuple[Any, dict[str, Any]]:
    """
    Peel the wrappers the reference puts around a derived-point function.

    Returns ``(named_function, keywords)``.  Handles ``functools.partial`` and the axle's
    side-qualifying closure (reference ``axle/suspension.py:273-282``), whose cell
    contents are the corner function and the side.
    """
    keywords: dict[str, Any] = {}
    seen = 0
    while seen < 8:
        seen += 1
        if isinstance(function, functools.partial):
            keywords = {**function.keywords, **keywords}
            function = function.func
            continue
        inner = getattr(function, "__okx_inner__", None)
        if inner is None:
            closure = getattr(function, "__closure__", None)
            if closure:
                for cell in closure:
                    try:
                        content = cell.cell_contents
                    except ValueError:
                        continue
                    if callable(content):
                        inner = content
                        break
        if inner is not None and inner is not function:
            function = inner
            continue
        break
    return function, keywords


def _flatten_derived(
    spec: Any, order: Sequence[Any], index: Mapping[Any, int]
) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Derived-point ops in evaluation order (reference ``manager.py:146-184``)."""
    dop_type, dop_out, dop_pts, dop_param = [], [], [], []
    for key in order:
        function, keywords = _unwrap_derived(spec.functions[key])
        deps = list(spec.dependencies[key])
        by_name = {_base_name(d): d for d in deps}

        def dep(name: str) -> int:
            try:
                return index[by_name[name]]
            except KeyError as error:
                raise ValueError(
                    f"derived point {key!r}: dependency {name} not declared"
                ) from error

        fname = getattr(function, "__name__", repr(function))
        pts = [-1, -1, -1, -1]
        if fname == "get_axle_midpoint":
            # definitions.py:76-89
            kind, param = DOP_MIDPOINT, 0.0
            pts[0], pts[1] = dep("AXLE_INBOARD"), dep("AXLE_OUTBOARD")
        elif fname == "get_wheel_center":
            # definitions.py:92-115: p1 - normalize(p1 - p2) * offset
            kind, param = DOP_ALONG, -float(keywords["wheel_offset"])
            pts[0] = pts[1] = dep("AXLE_OUTBOARD")
            pts[2] = dep("AXLE_INBOARD")
        elif fname == "get_wheel_inboard":
            # definitions.py:118-135: wc - normalize(wc - axi) * (width / 2)
            kind, param = DOP_ALONG, -(float(keywords["wheel_width"]) / 2)
            pts[0] = pts[1] = dep("WHEEL_CENTER")
            pts[2] = dep("AXLE_INBOARD")
        elif fname == "get_wheel_outboard":
            # definitions.py:138-155: wc + normalize(wc - axi) * (width / 2)
            kind, param = DOP_ALONG, float(keywords["wheel_width"]) / 2
            pts[0] = pts[1] = dep("WHEEL_CENTER")
            pts[2] = dep("AXLE_INBOARD")
        elif fname == "get_contact_patch_center":
            # definitions.py:158-180
            kind, param = DOP_CONTACT_PATCH, float(keywords["tire_radius"])
            pts[0], pts[1], pts[2] = dep("WHEEL_CENTER"), dep("AXLE_INBOARD"), dep("AXLE_OUTBOARD")
        elif fname == "get_point_along_line":
            # definitions.py:24-33: start + normalize(end - start) * d
            kind, param = DOP_ALONG, float(keywords["distance_from_start"])
            start = dep(_base_name(keywords["start_point"]))
            end = dep(_base_name(keywords["end_point"]))
            pts[0], pts[1], pts[2] = start, end, start
        else:
            raise NotImplementedError(
                f"derived point {key!r}: function {fname!r} has no device implementation"
            )
        dop_type.append(kind)
        dop_out.append(index[key])
        dop_pts.append(pts)
        dop_param.append(param)
    return (
        np.asarray(dop_type, dtype=np.int32).reshape(-1),
        np.asarray(dop_out, dtype=np.int32).reshape(-1),
        np.asarray(dop_pts, dtype=np.int32).reshape(-1, 4),
        np.asarray(dop_param, dtype=np.float64).reshape(-1),
    )


def derived_update_order(spec: Any) -> list[Any]:
    """Topological order with cycle detection (reference ``manager.py:146-184``)."""
    functions = spec.functions
    dependencies = spec.dependencies
    state: dict[Any, int] = {}
    order: list[Any] = []

    def visit(node: Any) -> None:
        mark = state.get(node, 0)
        if mark == 2:
            return
        if mark == 1:
            raise ValueError("Circular dependency detected in derived point definitions.")
        state[node] = 1
        for d in dependencies.get(node, ()):
            if d in functions:
                visit(d)
        state[node] = 2
        order.append(node)

    for key in functions:
        visit(key)
    return order


# --------------------------------------------------------------------------------------
# Constraints
# --------------------------------------------------------------------------------------


def describe_constraint(constraint: Any) -> str:
    """Reference ``solver.py:630-637``."""
    names = ", ".join(sorted(getattr(p, "name", str(p)) for p in constraint.involved_points))
    return f"{type(constraint).__name__}({names})"


def _flatten_constraint(
    c: Any, index: Mapping[Any, int], line_mode: str
) -> list[tuple[int, list[int], list[float]]]:
    name = type(c).__name__
    q = [0.0] * ROW_PARAMS

    def pts(*keys: Any) -> list[int]:
        out = [index[k] for k in keys]
        return out + [-1] * (ROW_POINTS - len(out))

    if name == "DistanceConstraint":
        q[0] = float(c.target_distance)
        return [(ROW_DISTANCE, pts(c.p1, c.p2), q)]
    if name == "SphericalJointConstraint":
        return [(ROW_SPHERICAL, pts(c.p1, c.p2), q)]
    if name == "AngleConstraint":
        q[0] = float(c.target_angle)
        return [(ROW_ANGLE, pts(c.v1_start, c.v1_end, c.v2_start, c.v2_end), q)]
    if name == "ThreePointAngleConstraint":
        q[0] = float(c.target_angle)
        return [(ROW_THREE_POINT_ANGLE, pts(c.p1, c.p2, c.p3), q)]
    if name == "VectorsParallelConstraint":
        return [(ROW_VECTORS_PARALLEL, pts(c.v1_start, c.v1_end, c.v2_start, c.v2_end), q)]
    if name == "VectorsPerpendicularConstraint":
        return [(ROW_VECTORS_PERPENDICULAR, pts(c.v1_start, c.v1_end, c.v2_start, c.v2_end), q)]
    if name == "EqualDistanceConstraint":
        return [(ROW_EQUAL_DISTANCE, pts(c.p1, c.p2, c.p3, c.p4), q)]
    if name == "FixedAxisConstraint":
        q[0] = float(int(getattr(c.axis, "value", c.axis)))
        q[1] = float(c.value)
        return [(ROW_FIXED_AXIS, pts(c.point_id), q)]
    if name == "PointOnLineConstraint":
        q[0:3] = _raw(c.line_point).tolist()
        q[3:6] = _raw(c.line_direction).tolist()
        if line_mode == "pinned":
            rows = []
            for comp in range(3):
                qq = list(q)
                qq[6] = float(comp)
                rows.append((ROW_LINE_PIN, pts(c.point_id), qq))
            return rows
        return [(ROW_POINT_ON_LINE, pts(c.point_id), q)]
    if name == "PointOnPlaneConstraint":
        q[0:3] = _raw(c.plane_point).tolist()
        q[3:6] = _raw(c.plane_normal).tolist()
        return [(ROW_POINT_ON_PLANE, pts(c.point_id), q)]
    if name == "MidpointOnPlaneConstraint":
        q[0:3] = _raw(c.plane_point).tolist()
        q[3:6] = _raw(c.plane_normal).tolist()
        return [(ROW_MIDPOINT_ON_PLANE, pts(c.point_a, c.point_b), q)]
    if name == "ScalarTripleProductConstraint":
        q[0] = float(c.target_volume)
        q[1] = float(c.scale)
        return [(ROW_SCALAR_TRIPLE, pts(c.p1, c.p2, c.p3, c.p4), q)]
    if name == "CoplanarPointsConstraint":
        return [(ROW_COPLANAR, pts(c.p1, c.p2, c.p3, c.p4), q)]
    raise TypeError(f"No device implementation for {name}")


def resolve_direction(direction: Any) -> np.ndarray:
    """Reference ``targeting.py:135-148`` without importing its types."""
    if hasattr(direction, "axis"):
        unit = np.zeros(3, dtype=np.float64)
        unit[int(getattr(direction.axis, "value", direction.axis))] = 1.0
        return unit
    if hasattr(direction, "vector"):
        return _raw(direction.vector).copy()
    raise TypeError(f"Unsupported target type: {type(direction)!r}")


def flatten_problem(
    initial_state: Any,
    constraints: Sequence[Any],
    derived_spec: Any,
    targets: Iterable[tuple[Any, Any]] = (),
    output_points: Sequence[Any] | None = None,
    line_mode: str = "softnorm",
) -> ConstraintProgram:
    """
    Flatten the arguments of the reference's ``solve_suspension_sweep`` into arrays.

    Args:
        initial_state: ``SuspensionState``-like (``positions`` dict incl. derived points,
            ``free_points`` set).
        constraints: constraint objects (reference classes or same-named lookalikes).
        derived_spec: ``DerivedPointsSpec``-like (``functions``, ``dependencies``).
        targets: one ``(point_key, direction)`` per sweep dimension.
        output_points: points written per solve (default: every point, state order).
        line_mode: ``"softnorm"`` keeps ``PointOnLineConstraint`` as the reference's
            scalar row; ``"pinned"`` expands it to three linear rows (``okx.h``).
    """
    if line_mode not in ("softnorm", "pinned"):
        raise ValueError("line_mode must be 'softnorm' or 'pinned'")
    positions = initial_state.positions
    free_sorted = sorted(initial_state.free_points)  # state.py:46-50
    derived_keys = list(derived_spec.functions.keys()) if derived_spec is not None else []

    point_keys = list(positions.keys())
    for key in derived_keys:
        if key not in positions:
            point_keys.append(key)
    index = {k: i for i, k in enumerate(point_keys)}

    role = np.zeros(len(point_keys), dtype=np.int8)
    for k in free_sorted:
        role[index[k]] = ROLE_FREE
    for k in derived_keys:
        if role[index[k]] == ROLE_FREE:
            raise ValueError(f"point {k!r} is both free and derived")
        role[index[k]] = ROLE_DERIVED

    order = derived_update_order(derived_spec) if derived_spec is not None else []
    dop_type, dop_out, dop_pts, dop_param = _flatten_derived(derived_spec, order, index)

    design = np.zeros((len(point_keys), 3), dtype=np.float64)
    for k, i in index.items():
        if k in positions:
            design[i] = _raw(positions[k])

    rows: list[tuple[int, list[int], list[float]]] = []
    source: list[int] = []
    desc: list[str] = []
    for ci, c in enumerate(constraints):
        flat = _flatten_constraint(c, index, line_mode)
        rows.extend(flat)
        source.extend([ci] * len(flat))
        desc.append(describe_constraint(c))

    tgt_point, tgt_dir, tgt_desc = [], [], []
    for key, direction in targets:
        if key not in index:
            raise ValueError(f"target point {key!r} is not part of the suspension state")
        if role[index[key]] == ROLE_FIXED:
            raise ValueError(f"target point {key!r} is fixed")
        tgt_point.append(index[key])
        unit = direction if isinstance(direction, np.ndarray) else resolve_direction(direction)
        tgt_dir.append(np.asarray(unit, dtype=np.float64))
        tgt_desc.append(
            f"target on point '{getattr(key, 'name', str(key))}' (direction {direction})"
        )

    if output_points is None:
        output_points = point_keys
    out_point = np.asarray([index[k] for k in output_points], dtype=np.int32)

    program = ConstraintProgram(
        point_keys=point_keys,
        role=role,
        free_point=np.asarray([index[k] for k in free_sorted], dtype=np.int32),
        dop_type=dop_type,
        dop_out=dop_out,
        dop_pts=dop_pts,
        dop_param=dop_param,
        row_type=np.asarray([r[0] for r in rows], dtype=np.int32).reshape(-1),
        row_pts=np.asarray([r[1] for r in rows], dtype=np.int32).reshape(-1, ROW_POINTS),
        row_param=np.asarray([r[2] for r in rows], dtype=np.float64).reshape(-1, ROW_PARAMS),
        row_source=np.asarray(source, dtype=np.int32).reshape(-1),
        tgt_point=np.asarray(tgt_point, dtype=np.int32).reshape(-1),
        tgt_dir=np.asarray(tgt_dir, dtype=np.float64).reshape(-1, 3),
        out_point=out_point,
        design_pos=design,
        constraint_desc=desc,
        target_desc=tgt_desc,
        line_mode=line_mode,
    )
    return program
