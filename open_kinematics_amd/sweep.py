"""
Drop-in for ``kinematics.core.sweep.solve_sweep`` (reference ``core/sweep.py:35-65``) plus the
array-level entry points that large batches should use.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .derived import DerivedPointsManager
from .program import ConstraintProgram, flatten_problem
from .solver import SolverConfig, absolute_target_table, solve_suspension_sweep
from .targeting import validate_sweep_controls


def solve_sweep(suspension, sweep_config, solver_config: SolverConfig = SolverConfig(), *, device=None):
    """
    Validate the sweep controls, then solve every step on the GPU.

    ``suspension`` is any object with the reference's ``Suspension`` protocol
    (``initial_state()``, ``constraints()``, ``derived_spec()``, ``actuator_dofs()``): the
    reference's own models work unchanged.
    """
    validate_sweep_controls(sweep_config, suspension.actuator_dofs())
    state, flattened = _dropin_flattened(suspension, sweep_config, solver_config)
    return solve_suspension_sweep(
        initial_state=state,
        constraints=None if flattened is not None else suspension.constraints(),
        sweep_config=sweep_config,
        derived_manager=None if flattened is not None else suspension.derived_spec(),
        solver_config=solver_config,
        device=device,
        _flattened=flattened,
    )


def _dropin_flattened(suspension, sweep_config, solver_config):
    """``(initial state, (program, table) or None)``: the drop-in's program - every point of a state an output, ``solver.py:763`` -
    from the suspension's memo (``solver.memoized_program``); None for sweeps that are solved run by run or have no step."""
    from .solver import (_coerce_config, memoized_program, target_segments, validate_least_squares_dimensions)

    if sweep_config.n_steps == 0 or len(target_segments(sweep_config)) > 1:
        return suspension.initial_state(), None
    cfg = _coerce_config(solver_config)

    def build(state, heads):
        constraints = suspension.constraints()
        validate_least_squares_dimensions(3 * len(state.free_points), len(constraints) + len(heads))
        spec = suspension.derived_spec()
        return flatten_problem(state, constraints, getattr(spec, "spec", spec), heads, output_points=None,
                               line_mode="softnorm").with_line_mode(cfg.line_mode)

    program, table, state = memoized_program(suspension, sweep_config, "dropin", cfg.line_mode, build, one_run=True)
    return state, (program, table)


def sweep_program(suspension, sweep_config, line_mode: str = "pinned") -> tuple[ConstraintProgram, np.ndarray]:
    """Constraint program (outputs = ``suspension.output_points()``) and absolute targets ``[S, T]``."""
    from .solver import memoized_program

    validate_sweep_controls(sweep_config, suspension.actuator_dofs())

    def build(state, heads):
        return flatten_problem(state, suspension.constraints(), suspension.derived_spec(), heads,
                               suspension.output_points(), line_mode="softnorm").with_line_mode(line_mode)

    program, table, _ = memoized_program(suspension, sweep_config, "sweep", line_mode, build)
    return program, table


def target_rows(suspension, specs, line_mode: str = "pinned") -> tuple[ConstraintProgram, np.ndarray]:
    """
    Program for explicit target rows ``[(point_key, direction), ...]``; also returns the
    design coordinate of every target so callers can build absolute grids on the device.
    """
    state = suspension.initial_state()
    program = flatten_problem(state, suspension.constraints(), suspension.derived_spec(), specs,
                              suspension.output_points(), line_mode="softnorm").with_line_mode(line_mode)
    base = np.array([float(np.dot(program.design_pos[p], d)) for p, d in zip(program.tgt_point, program.tgt_dir)])
    return program, base


# --------------------------------------------------------------------------------------
# metrics of a solved sweep (reference core/sweep.py:78-173, metrics/main.py:63-185)
# --------------------------------------------------------------------------------------


class SweepMetricsResult:
    """``core/sweep.py:78-87``: metric rows plus the derivative-computation status."""

    def __init__(self, rows, derivative_error=None, tangent_solve_infos=None):
        self.rows = rows
        self.derivative_error = derivative_error
        self.tangent_solve_infos = tangent_solve_infos


class AxleMetricRows:
    """``metrics/main.py:38-49``: axle-scope row plus one row per corner."""

    def __init__(self, axle, corners):
        self.axle = axle
        self.corners = corners

    def flat_row(self):
        """``metrics/main.py:52-62``: corner keys get a ``_left`` / ``_right`` suffix, axle keys follow."""
        from collections import OrderedDict

        flat = OrderedDict()
        for side, row in self.corners.items():
            for key, value in row.items():
                flat[f"{key}_{side.name.lower()}"] = value
        flat.update(self.axle)
        return flat


# (response, driver) of the derivative columns every corner declares (metrics/catalog.py:160-308), in the reference's order
_CORNER_DERIVATIVES = (
    ("camber", "hub_z"), ("roadwheel_angle", "hub_z"), ("caster", "hub_z"), ("kpi", "hub_z"), ("half_track", "hub_z"),
    ("wheel_center_x", "hub_z"), ("roadwheel_angle", "rack_displacement"), ("camber", "rack_displacement"),
)


EPS_GEOMETRIC = 1e-6  # primitives/constants.py:9


def _none_if_nan(value: float):
    return None if value != value else float(value)


def _driver_ratio(response_rate, driver_rate, candidates, column: str = ""):
    """
    ``DerivativeMetricDefinition.select_tangent`` + ``evaluate`` (``metrics/derivatives.py:265-320``) for a whole
    batch: along the candidate target whose tangent moves the driver coordinate most, response rate / driver rate.
    ``response_rate [B, T]``, ``driver_rate [B, T]`` (the driver coordinate's velocity along every target's tangent:
    ``tangents[:, :, point, axis]``, or the evaluated solve's rate columns) -> ``[B]`` (NaN where no candidate drives
    it).  Two candidates of equal strength (within ``EPS_GEOMETRIC``) are the reference's "Ambiguous derivative driver"
    ``ValueError`` (``derivatives.py:296-305``).
    """
    import numpy as np

    if isinstance(response_rate, np.ndarray):  # host arrays (a sweep's rows: a handful of values - no launches for them)
        b = response_rate.shape[0]
        if not candidates or driver_rate is None:
            return np.full((b,), np.nan)
        if len(candidates) == 1:  # the usual case (one hub target, one rack target): nothing to choose between
            rate, resp = driver_rate[:, candidates[0]], response_rate[:, candidates[0]]
            with np.errstate(divide="ignore", invalid="ignore"):
                return np.where(np.abs(rate) >= EPS_GEOMETRIC, resp / rate, np.nan)
        rates = driver_rate[:, candidates]                               # [B, C]
        strength = np.abs(rates)
        pick = strength.argmax(axis=1)[:, None]
        rate = np.take_along_axis(rates, pick, axis=1)[:, 0]
        if len(candidates) > 1:
            best = np.take_along_axis(strength, pick, axis=1)
            tied = (np.abs(best - strength) <= EPS_GEOMETRIC) & (strength >= EPS_GEOMETRIC)
            if bool((tied.sum(axis=1) > 1).any()):
                raise ValueError(f"Ambiguous derivative driver for column '{column}': "
                                 "multiple matching tangents have equal strength")
        resp = np.take_along_axis(response_rate[:, candidates], pick, axis=1)[:, 0]
        with np.errstate(divide="ignore", invalid="ignore"):
            out = resp / rate
        return np.where(np.abs(rate) >= EPS_GEOMETRIC, out, np.nan)

    import torch

    b = response_rate.shape[0]
    if not candidates or driver_rate is None:
        return torch.full((b,), float("nan"), dtype=torch.float64, device=response_rate.device)
    cand = torch.as_tensor(candidates, device=response_rate.device)
    rates = driver_rate[:, cand]                                      # [B, C]
    strength = rates.abs()
    pick = strength.argmax(dim=1, keepdim=True)                       # strongest driver rate
    rate = rates.gather(1, pick).squeeze(1)
    if len(candidates) > 1:
        best = strength.gather(1, pick)
        tied = ((best - strength).abs() <= EPS_GEOMETRIC) & (strength >= EPS_GEOMETRIC)
        if bool((tied.sum(dim=1) > 1).any()):
            raise ValueError(f"Ambiguous derivative driver for column '{column}': "
                             "multiple matching tangents have equal strength")
    resp = response_rate[:, cand].gather(1, pick).squeeze(1)
    out = resp / rate
    return torch.where(rate.abs() >= EPS_GEOMETRIC, out, torch.full_like(out, float("nan")))


def _corner_rows(corner, program, positions, tangents, side=None, rotation=None, actuators=(), evaluated=None):
    """
    Catalog + topology + derivative columns of one corner for every state -> list of OrderedDict.
    ``evaluated``: an ``EvaluatedResult`` (``DeviceProgram.evaluate`` / ``solve_evaluated``) that already holds this
    corner's metric values, derivatives and driver rates - nothing is launched then, and ``tangents`` may be None.
    """
    from collections import OrderedDict

    from .enums import PointID, PointRef
    from .metrics import CATALOG_ORDER, METRIC_NAMES, corner_roles, corner_state_metrics

    key = (lambda p: PointRef(side, p)) if side is not None else (lambda p: p)
    out_keys = [program.point_keys[k] for k in program.out_point]
    rack_point = corner.rack_attachment_point()
    rack_idx = out_keys.index(key(rack_point)) if rack_point is not None else -1
    if evaluated is not None:
        from ._abi import EVAL_RATE_RACK_Y, EVAL_RATE_WHEEL_CENTER_X

        # [B, 1 + T, 24]: ONE copy, the ratios below are host arithmetic (an axle's caller hands over its corner's block)
        block = evaluated if isinstance(evaluated, np.ndarray) else evaluated.eval.cpu().numpy()
        values = block[:, 0, : len(METRIC_NAMES)]
        derivatives = block[:, 1:, : len(METRIC_NAMES)]
        wc_rates = block[:, 1:, EVAL_RATE_WHEEL_CENTER_X : EVAL_RATE_WHEEL_CENTER_X + 3]
        rack_rates = block[:, 1:, EVAL_RATE_RACK_Y] if rack_point is not None else None
        have_rates = True
    else:
        roles = corner_roles(corner, program, side)
        res = corner_state_metrics(roles, positions, tangents)
        values = res.values.cpu().numpy()
        have_rates = tangents is not None
        derivatives = res.derivatives.cpu().numpy() if have_rates else None
        wc_rates = tangents[:, :, out_keys.index(key(PointID.WHEEL_CENTER)), :].cpu().numpy() if have_rates else None
        rack_rates = tangents[:, :, rack_idx, 1].cpu().numpy() if have_rates and rack_point is not None else None
    columns: "OrderedDict[str, Any]" = OrderedDict((n, values[:, METRIC_NAMES.index(n)]) for n in CATALOG_ORDER)
    rot_names, rot_values, rot_derivs = rotation if rotation is not None else ([], None, None)
    if rot_names:
        rot_host = _host(rot_values)
        for k, name in enumerate(rot_names):
            columns[name] = rot_host[:, k]
        rot_derivs = _host(rot_derivs) if have_rates and rot_derivs is not None else rot_derivs
    if have_rates:
        tgt_keys = [program.point_keys[p] for p in program.tgt_point]
        hub = [t for t, k in enumerate(tgt_keys) if k == key(PointID.WHEEL_CENTER)]
        # a shared actuator (the axle's rack) drives both corners: its targets count for either side
        # (metrics/main.py:103-148 _corner_tangents / _local_tangent_target)
        rack_keys = {key(rack_point)} if rack_point is not None else set()
        for actuator in actuators:
            if rack_keys & set(actuator.point_keys):
                rack_keys |= set(actuator.point_keys)
        rack = [t for t, k in enumerate(tgt_keys) if k in rack_keys]
        drivers = {"hub_z": (wc_rates[:, :, 2], hub), "rack_displacement": (rack_rates, rack)}

        def add(response: str, driver: str, rate):
            driver_rate, cand = drivers[driver]
            column = f"deriv_{response}_wrt_{driver}"
            columns[column] = _driver_ratio(rate, driver_rate, cand, column)

        for response, driver in _CORNER_DERIVATIVES:
            if driver == "rack_displacement" and rack_point is None:
                continue  # catalog.py: rack-driven derivatives are omitted without a steering rack
            rate = wc_rates[:, :, 0] if response == "wheel_center_x" else derivatives[:, :, METRIC_NAMES.index(response)]
            add(response, driver, rate)
        # topology declarations (corner/double_wishbone.py, macpherson.py:224-245): actuation first, then the spring
        for k, name in enumerate(rot_names):
            if name == "rocker_angle":
                add(name, "hub_z", rot_derivs[:, :, k])
        if corner.damper_points() is not None:
            add("damper_length", "hub_z", derivatives[:, :, METRIC_NAMES.index("damper_length")])
        for k, name in enumerate(rot_names):
            if name != "rocker_angle":
                add(name, "hub_z", rot_derivs[:, :, k])
    return _rows_from_columns(columns)


def _host(array):
    """A device tensor's host copy; host arrays pass through."""
    return array if isinstance(array, np.ndarray) else array.cpu().numpy()


def _rows_from_columns(columns) -> list:
    """``{name: values [n]}`` -> one OrderedDict per state, NaN read as None (the reference's "not defined here")."""
    from collections import OrderedDict

    import numpy as np

    names = list(columns)
    if not names:
        return []
    table = np.stack([np.asarray(columns[name], dtype=np.float64) for name in names], axis=1)   # [n, columns]
    cells = table.astype(object)
    cells[np.isnan(table)] = None
    return [OrderedDict(zip(names, row)) for row in cells.tolist()]


def compute_sweep_metrics(suspension, sweep_config, states, *, device=None) -> SweepMetricsResult:
    """
    Drop-in for ``kinematics.core.sweep.compute_sweep_metrics`` (``core/sweep.py:144-173``): every state's metric
    row — the corner catalog, the topology's extras, the derivative columns — computed on the device from the solved
    states (tangents -> metrics in HBM; the rows are built on the host at the end).  Corner suspensions give one
    ``OrderedDict`` per state, axles an ``AxleMetricRows`` (axle row + one row per side, ``flat_row()``).
    """
    from collections import OrderedDict

    import torch

    from .enums import Side
    from .metrics import AXLE_METRIC_NAMES, axis_rotation_metrics, axle_roles, axle_state_metrics, topology_rotation_roles
    from .sensitivity import _positions_array, solve_infos_from_records
    from .solver import _device_program

    if getattr(suspension, "config", True) is None:  # core/sweep.py:150-151
        return SweepMetricsResult(rows=[OrderedDict() for _ in states])
    program, _ = sweep_program(suspension, sweep_config)
    dp = _device_program(program, device)
    derivative_error = None
    tangent_infos = None
    out_keys = [program.point_keys[k] for k in program.out_point]
    positions = torch.as_tensor(_positions_array(states, out_keys), device=dp.device)
    tangents = None
    is_axle = hasattr(suspension, "corners")
    names, roles = topology_rotation_roles(suspension, program)
    if not is_axle and program.n_targets > 0:
        # a corner: tangents -> catalog -> derivatives in ONE launch (okx_evaluate_batch, the solve kernels' evaluation
        # epilogue on given states); the tangents themselves are only materialised for a topology's rotation metrics
        evaluated = _evaluate_states(dp, suspension, program, positions, want_tangents=bool(names))
        if evaluated is not None:
            from .sensitivity import solve_infos_from_records

            rot_values = rot_derivs = None
            if names:
                rot_values, rot_derivs = axis_rotation_metrics(roles, positions, evaluated.tangents)
            rows = _corner_rows(suspension, program, positions, evaluated.tangents, None, (names, rot_values, rot_derivs),
                                evaluated=evaluated)
            return SweepMetricsResult(rows, None, solve_infos_from_records(evaluated.tangent_info(), program.n_vars))
    if is_axle and program.n_targets > 0:
        # a composed axle: tangents, both corners' catalogs, the axle-scope metrics and the rotation / hardware roles in ONE
        # launch on the given states (the pair-mode evaluated module)
        enabled = _enable_axle_evaluation(dp, suspension, program)
        if enabled is not None:
            evaluated = dp.evaluate(positions)
            rows = _axle_rows_from_evaluated(suspension, program, evaluated, *enabled)
            return SweepMetricsResult(rows, None, solve_infos_from_records(evaluated.tangent_info(), program.n_vars))
    if program.n_targets > 0:
        try:
            tangents, tinfo = dp.tangents(positions)
            tangent_infos = solve_infos_from_records(dp.tangent_info(tinfo), program.n_vars)
        except Exception as error:  # noqa: BLE001 - metrics degrade without derivatives (core/sweep.py:155-160)
            derivative_error = f"{type(error).__name__}: {error}"
    rot_values = rot_derivs = None
    if names:
        rot_values, rot_derivs = axis_rotation_metrics(roles, positions, tangents)
    if not is_axle:
        rows = _corner_rows(suspension, program, positions, tangents, None, (names, rot_values, rot_derivs))
        return SweepMetricsResult(rows, derivative_error, tangent_infos)
    per_side = {}
    for side in (Side.LEFT, Side.RIGHT):
        tag = side.name.lower()
        mine = [k for k, n in enumerate(names) if n.endswith("_" + tag) and not n.startswith("arb_arm_angle")]
        sub = ([names[k][: -len(tag) - 1] for k in mine],
               None if rot_values is None else rot_values[:, mine],
               None if rot_derivs is None else rot_derivs[:, :, mine])
        per_side[side] = _corner_rows(suspension.corners[side], program, positions, tangents, side, sub,
                                      suspension.actuator_dofs())
    left, right = axle_roles(suspension, program)
    axle_values = axle_state_metrics(left, right, positions).cpu().numpy()
    from .enums import PointID, PointRef
    from .metrics import axle_hardware_metrics

    hub_rates = {side: tangents[:, :, out_keys.index(PointRef(side, PointID.WHEEL_CENTER)), 2].cpu().numpy()
                 for side in (Side.LEFT, Side.RIGHT)} if tangents is not None else {}
    hardware = {k: _host(v) for k, v in axle_hardware_metrics(suspension, program, positions, tangents).items()}
    rows = _axle_rows(program, len(states), per_side, axle_values, names, None if rot_values is None else _host(rot_values),
                      None if rot_derivs is None else _host(rot_derivs), hardware, hub_rates)
    return SweepMetricsResult(rows, derivative_error, tangent_infos)


def _axle_rows(program, n_states: int, per_side: dict, axle_values, names, rot_values, rot_derivs, hardware: dict, hub_rates: dict) -> list:
    """
    ``AxleMetricRows`` per state from host arrays: the two corners' rows, the axle-scope metrics ``[B, 7]``, the rotation
    roles (``names`` / ``rot_values [B, K]`` / ``rot_derivs [B, T, K]`` of ``topology_rotation_roles``), the shared
    hardware's values and rates (``axle_hardware_metrics``' dictionary) and each side's wheel-centre z rate ``[B, T]``
    (empty: no derivative columns).  Order: ``axle/suspension.py:213-238``.
    """
    from collections import OrderedDict

    from .enums import PointID, PointRef, Side
    from .metrics import AXLE_METRIC_NAMES

    arm = {side: names.index(f"arb_arm_angle_{side.name.lower()}") for side in (Side.LEFT, Side.RIGHT)} \
        if "arb_arm_angle_left" in names else None
    tgt_keys = [program.point_keys[p] for p in program.tgt_point]
    rows = []
    for s in range(n_states):
        axle_row = OrderedDict((n, _none_if_nan(axle_values[s, k])) for k, n in enumerate(AXLE_METRIC_NAMES))
        rows.append(AxleMetricRows(axle_row, {side: per_side[side][s] for side in (Side.LEFT, Side.RIGHT)}))
    have_rates = bool(hub_rates)

    def hub_z_columns(response: str, rate) -> dict:
        """``deriv_<response>_wrt_hub_z_<side>`` per side from the response's rate along every target's tangent."""
        columns = OrderedDict()
        for side in (Side.LEFT, Side.RIGHT):
            key = PointRef(side, PointID.WHEEL_CENTER)
            cand = [t for t, k in enumerate(tgt_keys) if k == key]
            column = f"deriv_{response}_wrt_hub_z_{side.name.lower()}"
            columns[column] = _driver_ratio(rate, hub_rates[side], cand, column)
        return columns

    # the shared hardware's state metrics, then its derivative columns: anti-roll bar first, heave link second
    # (axle/suspension.py:213-238)
    state_columns: OrderedDict = OrderedDict()
    deriv_columns: OrderedDict = OrderedDict()
    if arm is not None:  # U-bar: arm angles about the bar's axis; the twist is their difference
        state_columns["arb_twist"] = rot_values[:, arm[Side.LEFT]] - rot_values[:, arm[Side.RIGHT]]
        if have_rates:
            deriv_columns.update(hub_z_columns("arb_twist", rot_derivs[:, :, arm[Side.LEFT]] - rot_derivs[:, :, arm[Side.RIGHT]]))
    if "t_bar_heave_angle" in hardware:  # rigid T-bar (axle/mechanisms.py:718-797)
        state_columns["t_bar_heave_angle"] = hardware["t_bar_heave_angle"]
        state_columns["arb_twist"] = hardware["arb_twist"]
        if have_rates:
            center, twist = hub_z_columns("t_bar_center_x", hardware["d_t_bar_center_x"]), hub_z_columns("arb_twist", hardware["d_arb_twist"])
            for side in ("left", "right"):  # per driver: centre travel, then twist
                for columns in (center, twist):
                    deriv_columns.update({k: v for k, v in columns.items() if k.endswith("_" + side)})
    if "heave_link_length" in hardware:  # mechanisms.py:903-944
        state_columns["heave_link_length"] = hardware["heave_link_length"]
        if have_rates:
            deriv_columns.update(hub_z_columns("heave_link_length", hardware["d_heave_link_length"]))
    arm_values = {side: rot_values[:, arm[side]] for side in arm} if arm is not None else {}
    for s, row in enumerate(rows):
        for name, col in (*state_columns.items(), *deriv_columns.items()):
            row.axle[name] = _none_if_nan(col[s])
        for side, col in arm_values.items():
            row.corners[side]["arb_arm_angle"] = _none_if_nan(col[s])
    return rows


def _axle_rows_from_evaluated(suspension, program, evaluated, roles, rot_names, hw_names) -> list:
    """The rows of ``compute_sweep_metrics`` for a composed axle from ONE evaluated launch (``okx.h`` OKX_EVAL_AXLE_*):
    both corners' blocks, the axle-scope metrics and the role columns of ``metrics.axle_evaluation_roles``."""
    from ._abi import (EVAL_AXLE_LEFT, EVAL_AXLE_METRICS, EVAL_AXLE_RIGHT, EVAL_AXLE_ROLES, EVAL_COLUMNS,
                       EVAL_RATE_WHEEL_CENTER_Z)
    from .enums import Side

    block = evaluated.eval.cpu().numpy()  # [B, 1 + T, 64]: ONE copy
    column = lambda name: EVAL_AXLE_ROLES + roles.column_of[name]  # noqa: E731
    rot_values = np.stack([block[:, 0, column(n)] for n in rot_names], axis=1) if rot_names else None
    rot_derivs = np.stack([block[:, 1:, column(n)] for n in rot_names], axis=2) if rot_names else None
    per_side, hub_rates = {}, {}
    for side, lo in ((Side.LEFT, EVAL_AXLE_LEFT), (Side.RIGHT, EVAL_AXLE_RIGHT)):
        tag = side.name.lower()
        mine = [k for k, n in enumerate(rot_names) if n.endswith("_" + tag) and not n.startswith("arb_arm_angle")]
        sub = ([rot_names[k][: -len(tag) - 1] for k in mine],
               None if rot_values is None else rot_values[:, mine], None if rot_derivs is None else rot_derivs[:, :, mine])
        corner_block = block[:, :, lo:lo + EVAL_COLUMNS]
        per_side[side] = _corner_rows(suspension.corners[side], program, None, None, side, sub, suspension.actuator_dofs(),
                                      evaluated=corner_block)
        hub_rates[side] = corner_block[:, 1:, EVAL_RATE_WHEEL_CENTER_Z]
    hardware = {}
    for n in hw_names:
        hardware[n] = block[:, 0, column(n)]
        hardware["d_" + n] = block[:, 1:, column(n)]
    return _axle_rows(program, block.shape[0], per_side, block[:, 0, EVAL_AXLE_METRICS:EVAL_AXLE_METRICS + 7], rot_names,
                      rot_values, rot_derivs, hardware, hub_rates)


def _enable_axle_evaluation(dp, suspension, program):
    """``(roles, rotation names, hardware names)`` once the program's axle evaluated kernels are loaded; None when it has
    none (no pair-mode kernel, more than eight roles, a role point outside the outputs, no compiler)."""
    from .metrics import axle_evaluation_roles

    try:
        roles, rot_names, hw_names = axle_evaluation_roles(suspension, program)
        dp.wait_ready()  # (the evaluated module is the generated pair kernel's: not while the interpreter still serves)
        dp.enable_evaluation(roles)
    except (ValueError, RuntimeError):
        return None
    return roles, rot_names, hw_names


def _evaluate_states(dp, suspension, program, positions, want_tangents: bool):
    """``DeviceProgram.evaluate`` for a corner whose program has evaluated kernels; None when it has none (the callers
    then run tangents and metrics as separate launches)."""
    from .metrics import corner_roles

    try:
        dp.enable_evaluation(corner_roles(suspension, program))
    except (ValueError, RuntimeError):  # no single-mode quad kernel, a role point outside the outputs, no compiler ...
        return None
    return dp.evaluate(positions, tangents=want_tangents)


# --------------------------------------------------------------------------------------
# the evaluated sweep (reference core/sweep.py:91-113, :217-270)
# --------------------------------------------------------------------------------------


class DerivativeIssue:
    """The advisory the reference attaches when the tangent computation failed or met a rank-deficient system
    (``core/sweep.py:176-214`` ``_derivative_issues``; its ``DiagnosticIssue`` fields, category ``derivatives``)."""

    category = "derivatives"
    severity = "warning"

    def __init__(self, step, message: str, value=None):
        self.step, self.message, self.value = step, message, value

    def __repr__(self) -> str:
        return f"DerivativeIssue(step={self.step!r}, message={self.message!r}, value={self.value!r})"


class EvaluatedSweep:
    """
    ``core/sweep.py:91-113``: solved states, solver statistics and metric rows of one sweep, the same length each.
    ``diagnostics`` carries the derivative advisories only: the reference's sweep diagnostics (``core/diagnostics.py``)
    are outside this package's scope.
    """

    def __init__(self, states, solver_stats, metrics, diagnostics):
        lengths = (len(states), len(solver_stats), len(metrics.rows))
        if len(set(lengths)) != 1:
            raise ValueError("Evaluated sweep state, solver-stat, and metric counts must match: "
                             f"{lengths[0]} states, {lengths[1]} solver stats, {lengths[2]} metric rows.")
        self.states, self.solver_stats, self.metrics, self.diagnostics = states, solver_stats, metrics, diagnostics


def _derivative_issues(result: SweepMetricsResult) -> list:
    """``core/sweep.py:176-214``."""
    issues = []
    if result.derivative_error is not None:
        issues.append(DerivativeIssue(None, "Derivative metrics unavailable: tangent computation failed "
                                            f"({result.derivative_error}); derivative columns are omitted."))
    infos = result.tangent_solve_infos or []
    deficient = [step for step, info in enumerate(infos) if info.rank_deficient]
    if deficient:
        first = deficient[0]
        min_sv = min(infos[step].smallest_singular_value for step in deficient)
        issues.append(DerivativeIssue(
            first, f"Tangent system rank-deficient at {len(deficient)} of {len(infos)} steps (first at step {first}, rank "
                   f"{infos[first].rank}/{infos[first].n_variables}, smallest singular value {min_sv:.3g}); derivative "
                   "values may not be unique.", min_sv))
    return issues


def evaluate_solved_sweep(suspension, sweep_config, states, solver_stats, *, device=None) -> EvaluatedSweep:
    """
    Drop-in for ``kinematics.core.sweep.evaluate_solved_sweep`` (``core/sweep.py:217-245``): metric rows and the
    derivative advisories of an already solved sweep.  For a corner the tangents, the catalog and its derivative columns
    are ONE kernel launch on the given states (``okx_evaluate_batch``).
    """
    if len(states) != len(solver_stats):
        raise ValueError(f"Solved state and solver-stat counts must match: {len(states)} states, {len(solver_stats)} solver stats.")
    metrics = compute_sweep_metrics(suspension, sweep_config, states, device=device)
    return EvaluatedSweep(states, solver_stats, metrics, _derivative_issues(metrics))


def solve_evaluated_sweep(suspension, sweep_config, solver_config: SolverConfig = SolverConfig(), *, device=None,
                          fused: bool | None = None) -> EvaluatedSweep:
    """
    Drop-in for ``kinematics.core.sweep.solve_evaluated_sweep`` (``core/sweep.py:248-270``): solve one sweep and compute
    its metric rows.  For a corner whose program has evaluated kernels the whole of it - every step's solve, its
    solution-manifold tangents, the metric catalog and the derivative columns - is ONE kernel launch
    (``okx_solve_evaluated_batch``: the tangents and metrics are the solve kernel's epilogue, taken at the converged state
    while it is still in registers); so is a composed axle's (both corners' catalogs, the axle-scope metrics and the
    rotation / hardware roles: the pair-mode evaluated module); programs without such kernels solve first and evaluate
    after (``evaluate_solved_sweep``).  ``fused``: see ``solver.solve_suspension_sweep(evaluation_fused=)`` - by default
    independent cold starts (``warm_start=False``, and the cold starts side by side a warm-started sweep of four steps and
    more is first solved as) take the one fused launch; a warm-started CHAIN (shorter sweeps, the fallback when the cold
    starts are not the sequential path) is solved as such and then evaluated by ONE more launch on the records in HBM (every
    step side by side); no host round trip either way.  Same states, same error behaviour as ``solve_sweep``.
    """
    from .metrics import axis_rotation_metrics, axle_evaluation_roles, corner_roles, topology_rotation_roles
    from .sensitivity import solve_infos_from_records

    validate_sweep_controls(sweep_config, suspension.actuator_dofs())
    is_axle = hasattr(suspension, "corners")
    with_metrics = getattr(suspension, "config", True) is not None and sweep_config.n_steps > 0
    if with_metrics:
        axle_parts = []

        def roles_of(program):
            if is_axle:  # both corners' roles and the rotation / hardware roles: everything a row needs, no tangent tensor
                axle_parts[:] = axle_evaluation_roles(suspension, program)
                return axle_parts[0], False
            return corner_roles(suspension, program), bool(topology_rotation_roles(suspension, program)[0])

        state, flattened = _dropin_flattened(suspension, sweep_config, solver_config)
        states, stats, extra = solve_suspension_sweep(
            initial_state=state, constraints=None if flattened is not None else suspension.constraints(), sweep_config=sweep_config,
            derived_manager=None if flattened is not None else suspension.derived_spec(), solver_config=solver_config, device=device,
            evaluation=roles_of, evaluation_fused=fused, _flattened=flattened)
        if extra is not None and is_axle:
            program, evaluated = extra
            rows = _axle_rows_from_evaluated(suspension, program, evaluated, *axle_parts)
            metrics = SweepMetricsResult(rows, None, solve_infos_from_records(evaluated.tangent_info(), program.n_vars))
            return EvaluatedSweep(states, stats, metrics, _derivative_issues(metrics))
        if extra is not None:
            program, evaluated = extra
            names, roles = topology_rotation_roles(suspension, program)
            rot_values = rot_derivs = None
            positions = evaluated.positions
            if names:
                rot_values, rot_derivs = axis_rotation_metrics(roles, positions, evaluated.tangents)
            rows = _corner_rows(suspension, program, positions, evaluated.tangents, None, (names, rot_values, rot_derivs),
                                evaluated=evaluated)
            metrics = SweepMetricsResult(rows, None, solve_infos_from_records(evaluated.tangent_info(), program.n_vars))
            return EvaluatedSweep(states, stats, metrics, _derivative_issues(metrics))
    else:
        states, stats = solve_sweep(suspension, sweep_config, solver_config, device=device)
    return evaluate_solved_sweep(suspension, sweep_config, states, stats, device=device)
