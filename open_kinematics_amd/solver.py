"""
Drop-in for the reference's ``kinematics.core.solver`` entry points
(``solve_suspension_sweep``, ``SolverConfig``, ``SolverInfo``, ``convert_targets_to_absolute``)
on top of the device solver.  Inputs may be the reference's own objects (duck-typed) or
this package's; the returned states are built with the same classes as the input state.
"""

from __future__ import annotations

import hashlib
import threading
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, NamedTuple

import numpy as np

from .enums import TargetPositionMode
from .program import ConstraintProgram, flatten_problem, resolve_direction
from .state import Point3, SuspensionState

SOLVE_TOLERANCE_VALUE = 1e-5
SOLVE_TOLERANCE_STEP = 1e-9
SOLVE_TOLERANCE_GRAD = 1e-9
SOLVE_ACCEPT_RESIDUAL = 1e-3


class SolverConfig(NamedTuple):
    """
    Reference fields first (``solver.py:65-80``), then the device solver's own controls.

    ``ftol`` / ``xtol`` / ``gtol`` are MINPACK's stopping tests in the reference (``solver.py:158-169``).  At the
    reference's default values or tighter the device solver simply iterates to its own fixed point (``step_tol``), which
    satisfies all three and costs no extra pass; values LOOSER than the defaults are honoured, see ``device_tolerances``.
    """

    ftol: float = SOLVE_TOLERANCE_VALUE
    xtol: float = SOLVE_TOLERANCE_STEP
    gtol: float = SOLVE_TOLERANCE_GRAD
    verbose: int = 0
    residual_tolerance: float = SOLVE_ACCEPT_RESIDUAL
    # device-solver controls
    line_mode: str = "pinned"   # "pinned" | "softnorm" (include/okx.h, OKX_ROW_LINE_PIN)
    warm_start: bool = True     # reference semantics: step k starts from step k-1 (solver.py:774)
    step_tol: float = 1e-11     # mm
    max_iter: int = 100
    parallel_chains: bool = True  # warm-started sweeps of four steps and more: every step a cold start, side by side, kept when it is the sequential path


def device_tolerances(cfg: "SolverConfig", program: ConstraintProgram) -> dict:
    """
    ``okx_solve_opts`` stopping tolerances for a ``SolverConfig``.

    MINPACK stops on ``xtol`` when the step is at most ``xtol * ||x||`` (``lmder``: ``delta <= xtol * xnorm``, unit
    scaling) and on ``ftol`` when the actual and the predicted relative cost reductions of an accepted step are both at
    most ``ftol``.  The device solver has the same two tests (``step_tol`` in millimetres on the largest coordinate
    change, ``ftol`` relative).  A caller who loosens ``xtol`` / ``ftol`` beyond the reference's defaults gets them:
    ``step_tol = max(cfg.step_tol, xtol * ||x0||_2)`` with ``x0`` the free coordinates of the design state, and
    ``ftol`` passed through - the solve stops earlier, with fewer evaluations, like the reference's would.  Defaults and
    tighter values leave the device's fixed-point stop (``cfg.step_tol``, device ``ftol`` 1e-10) in charge.  A loosened
    ``gtol`` is passed on as MINPACK's own test - ``okx_solve_opts.grad_tol < 0``: stop when
    ``max_j |(J^T r)_j| / (|J_j| |r|) <= gtol`` (``lmder``'s ``gnorm``) - which, as in the reference, only ever ends a solve
    whose residual does not vanish (a compromise point beyond the reach); such launches run without the shared first step.
    """
    tolerances = {"step_tol": float(cfg.step_tol)}
    if cfg.xtol > SOLVE_TOLERANCE_STEP:
        x0 = np.asarray(program.design_pos, dtype=np.float64)[np.asarray(program.free_point, dtype=np.int64)]
        tolerances["step_tol"] = max(tolerances["step_tol"], float(cfg.xtol) * float(np.linalg.norm(x0)))
    if cfg.ftol > SOLVE_TOLERANCE_VALUE:
        tolerances["ftol"] = float(cfg.ftol)
    if cfg.gtol > SOLVE_TOLERANCE_GRAD:
        tolerances["grad_tol"] = -float(cfg.gtol)
    return tolerances


@dataclass
class SolverInfo:
    """``solver.py:83-96``; ``nfev`` counts device residual evaluations."""

    converged: bool
    nfev: int
    max_residual: float


def validate_least_squares_dimensions(n_vars: int, n_residuals: int, *, method: str = "lm") -> None:
    if method == "lm" and n_vars > n_residuals:
        raise ValueError(
            f"System is underdetermined (n_vars={n_vars} > m_res={n_residuals}). "
            "The solve method (Levenberg-Marquardt) requires at least as "
            "many residuals as variables."
        )


def _is_absolute(mode) -> bool:
    return str(getattr(mode, "value", mode)).lower() == "absolute"


def convert_targets_to_absolute(targets, initial_state):
    """Relative displacement -> absolute scalar along the direction (``solver.py:584-627``)."""
    resolved = []
    for target in targets:
        if _is_absolute(target.mode):
            resolved.append(target)
            continue
        unit = resolve_direction(target.direction)
        position = initial_state.positions[target.point_id]
        coordinate = float(np.dot(np.asarray(getattr(position, "data", position)), unit))
        resolved.append(type(target)(
            point_id=target.point_id, direction=target.direction,
            value=coordinate + target.value,
            mode=_absolute_like(target.mode),
        ))
    return resolved


def _absolute_like(mode):
    try:
        return type(mode)("absolute")
    except Exception:
        return TargetPositionMode.ABSOLUTE


def target_segments(sweep_config) -> list:
    """
    ``[(first step, last step + 1)]``: maximal runs of steps over which every sweep dimension keeps its target point and
    direction.  The reference pairs arbitrary ``PointTarget``s by index (``targeting.py:67-75``), so a dimension may
    change the point or the direction it drives from one step to the next; the batched solver shares its target ROWS
    across the steps of one launch, so such a sweep is solved run by run.
    """
    dims = sweep_config.target_sweeps
    n_steps = sweep_config.n_steps
    cuts = set()
    for dim in dims:
        prev, prev_unit = dim[0], None
        for k in range(1, n_steps):
            t = dim[k]
            # (a dimension built by build_sweep shares ONE direction object: the identity test settles nearly every step)
            same = t.point_id == prev.point_id and t.direction is prev.direction
            if not same and t.point_id == prev.point_id:
                if prev_unit is None:
                    prev_unit = resolve_direction(prev.direction)
                same = bool(np.array_equal(resolve_direction(t.direction), prev_unit))
            if not same:
                cuts.add(k)
                prev_unit = None
            prev = t
    bounds = [0] + sorted(cuts) + [n_steps]
    return [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]


def absolute_target_table(sweep_config, initial_state, steps: tuple | None = None) -> tuple[list, np.ndarray]:
    """``([(point, direction)] per dimension, values [S, T])`` with the reference's arithmetic, for the steps
    ``[steps[0], steps[1])`` (default: the whole sweep, which must then keep one point and direction per dimension)."""
    dims = sweep_config.target_sweeps
    lo, hi = steps if steps is not None else (0, sweep_config.n_steps)
    if steps is None and len(target_segments(sweep_config)) > 1:
        raise NotImplementedError(
            "a sweep dimension changes its target point or direction between steps: one launch shares its target rows "
            "across its steps (solve_suspension_sweep solves such a sweep run by run, see target_segments)"
        )
    heads, columns = [], []
    for dim in dims:
        first = dim[lo] if hi > lo else dim[0]
        unit0 = resolve_direction(first.direction)
        position = initial_state.positions[first.point_id]
        base = float(np.dot(np.asarray(getattr(position, "data", position)), unit0))
        mode0 = first.mode  # (a dimension's targets nearly always share ONE mode object: the identity test settles them)
        shift0 = 0.0 if _is_absolute(mode0) else base
        columns.append([shift0 + t.value if t.mode is mode0 else (t.value if _is_absolute(t.mode) else base + t.value)
                        for t in dim[lo:hi]])
        heads.append((first.point_id, first.direction))
    table = np.asarray(columns, dtype=np.float64).T.reshape(hi - lo, len(dims))
    return heads, np.ascontiguousarray(table)


def describe_worst_residual(program: ConstraintProgram, residuals: np.ndarray) -> str:
    """``solver.py:640-651`` on the flattened row table."""
    worst = int(np.argmax(np.abs(residuals)))
    if worst < program.n_rows:
        return f"constraint {program.constraint_desc[int(program.row_source[worst])]}"
    return program.target_desc[worst - program.n_rows]


# Device programs of recent drop-in calls, most recently used last.  A repeated ``solve_sweep`` on the same
# suspension (the reference's own benchmark, tests/benchmarks/test_bench_sweep.py:29-40, does exactly that)
# then skips the upload, the kernel generation and the code-object load; the key covers the structure AND the
# geometry values of the flattened program, the line mode and the device, so a hit is the same program.
_PROGRAM_CACHE: "OrderedDict[tuple, Any]" = OrderedDict()
_PROGRAM_CACHE_LOCK = threading.Lock()
PROGRAM_CACHE_SIZE = 8


def _program_digest(program: ConstraintProgram) -> str:
    digest = hashlib.blake2b(digest_size=16)
    for array in (program.role, program.free_point, program.dop_type, program.dop_out, program.dop_pts,
                  program.dop_param, program.row_type, program.row_pts, program.row_param, program.tgt_point,
                  program.tgt_dir, program.out_point, program.design_pos):
        data = np.ascontiguousarray(array)
        digest.update(str(data.dtype).encode() + str(data.shape).encode())
        digest.update(data.tobytes())
    return digest.hexdigest()


def _program_key(program: ConstraintProgram, device) -> tuple:
    # (`_okx_digest`: left on the programs of `memoized_program` - this module's own objects, flattened once and never
    #  edited - so that a repeated sweep does not hash thirteen arrays per call; any other program is hashed here)
    digest = program.__dict__.get("_okx_digest") or _program_digest(program)
    return digest, program.line_mode, str(device)


def _device_program(program: ConstraintProgram, device=None):
    """Cached ``DeviceProgram`` of this exact program on this device (never closed by the callers)."""
    import torch

    from .batch import DeviceProgram

    if torch.cuda.is_available():
        device = torch.device(device if device is not None else "cuda")
        if device.type == "cuda" and device.index is None:  # 'cuda' and 'cuda:0' are the same cache entry
            device = torch.device("cuda", torch.cuda.current_device())
    key = _program_key(program, device)
    with _PROGRAM_CACHE_LOCK:
        dp = _PROGRAM_CACHE.get(key)
        if dp is not None:
            _PROGRAM_CACHE.move_to_end(key)
            return dp
    dp = DeviceProgram(program, device, wait_for_kernels=False)  # start solving at once; generated kernels take over when compiled
    with _PROGRAM_CACHE_LOCK:
        _PROGRAM_CACHE[key] = dp
        # an evicted program is only DROPPED: another caller (another thread, a BatchResult's owner) may still hold it,
        # and its device memory goes when the last reference does (DeviceProgram.__del__)
        while len(_PROGRAM_CACHE) > PROGRAM_CACHE_SIZE:
            _PROGRAM_CACHE.popitem(last=False)
    return dp


def clear_program_cache() -> None:
    """Forget every cached device program (each is released when its last holder lets go of it)."""
    with _PROGRAM_CACHE_LOCK:
        _PROGRAM_CACHE.clear()


def dropin_program(initial_state, constraints, sweep_config, derived_manager, solver_config=SolverConfig(),
                   output_points=None) -> tuple[ConstraintProgram, np.ndarray]:
    """The constraint program and absolute target table ``[S, T]`` that ``solve_suspension_sweep`` solves
    (``__graft_entry__.build()`` precompiles the kernels of the BASELINE sweeps through this)."""
    spec = getattr(derived_manager, "spec", derived_manager)
    cfg = _coerce_config(solver_config)
    heads, table = absolute_target_table(sweep_config, initial_state)
    n_vars = 3 * len(initial_state.free_points)
    validate_least_squares_dimensions(n_vars, len(constraints) + len(heads))
    program = flatten_problem(initial_state, constraints, spec, heads, output_points=output_points,
                              line_mode="softnorm").with_line_mode(cfg.line_mode)
    return program, table


PROGRAM_MEMO_SIZE = 8


def memoized_program(suspension, sweep_config, kind: str, line_mode: str, build, one_run: bool = False) -> tuple:
    """
    ``(program, absolute target table [S, T], initial state)`` of a suspension OBJECT and a one-run sweep, the flattened
    program taken from a small memo kept on the object when it has been flattened before.  Flattening re-emits every
    constraint (``Suspension.constraints()``) and walks the derived-point graph: 0.3 - 1 ms of host time per call, as
    much as the sweep's GPU work and more (``tools/dropin_profile.py``).  The key covers what a program depends on:
    the kind of output list, the line-row form, every design position of the initial state (bytes), and each
    dimension's target point and direction - so a suspension whose hardpoints were edited between calls flattens
    afresh.  (Edits that change the constraint SET without moving a point are not seen: build a new suspension, or
    ``clear_program_memo(suspension)``.)  ``build(state, heads) -> ConstraintProgram`` flattens on a miss.
    """
    state = suspension.initial_state()
    # (`one_run`: the caller has already seen that every dimension keeps its point and direction - target_segments is a
    #  Python loop over all steps, not worth running twice per call)
    heads, table = absolute_target_table(sweep_config, state, (0, sweep_config.n_steps) if one_run else None)
    store = getattr(suspension, "__dict__", None)
    if store is None:
        return build(state, heads), table, state
    points = state.positions
    fingerprint = hash(np.asarray([getattr(v, "data", v) for v in points.values()], dtype=np.float64).tobytes())
    key = (kind, line_mode, len(points), fingerprint,
           tuple((pid, resolve_direction(direction).tobytes()) for pid, direction in heads))
    with _PROGRAM_CACHE_LOCK:
        memo = store.setdefault("_okx_program_memo", OrderedDict())
        program = memo.get(key)
        if program is not None:
            memo.move_to_end(key)
    if program is None:
        program = build(state, heads)  # (outside the lock: flattening is the slow part; two threads may both flatten once)
        program.__dict__["_okx_digest"] = _program_digest(program)
        with _PROGRAM_CACHE_LOCK:
            memo[key] = program
            while len(memo) > PROGRAM_MEMO_SIZE:
                memo.popitem(last=False)
    return program, table, state


def clear_program_memo(suspension) -> None:
    """Forget the flattened programs remembered on this suspension object (``memoized_program``)."""
    getattr(suspension, "__dict__", {}).pop("_okx_program_memo", None)


def solve_suspension_sweep(initial_state, constraints, sweep_config, derived_manager,
                           solver_config=SolverConfig(), *, output_points=None, device=None, evaluation=None,
                           evaluation_fused: bool | None = None, _flattened=None):
    """
    Solve every step of a sweep on the GPU (reference ``solver.py:654-776``).

    Returns ``(states, infos)`` like the reference.  Raises ``ValueError`` for an
    underdetermined system and ``RuntimeError`` at the first step that did not converge or
    whose worst residual exceeds ``residual_tolerance`` — with the reference's messages.
    ``output_points`` (extension, default ``None`` = every point like the reference, ``solver.py:763``)
    restricts the returned states to those point keys.
    ``evaluation`` (extension; what ``sweep.solve_evaluated_sweep`` passes): a callable ``program -> (okx_corner_roles,
    want_tangents)``.  The launch that solves the sweep then also evaluates it (``okx_solve_evaluated_batch``: tangents
    and metrics as the solve kernel's epilogue) and a third value is returned: ``(program, EvaluatedResult)``, or ``None``
    when this sweep had to be solved without (no evaluated kernels for the program, target rows that change between steps).
    ``evaluation_fused``: True = that one launch; False = two launches, the solve and then ``okx_evaluate_batch`` on its
    records still in HBM - a warm-started sweep is a CHAIN, whose steps one quad walks one after the other, and the fused
    epilogue lengthens every step of it, while the given-states kernel evaluates all steps side by side: for the sweeps
    the drop-in sees (tens of steps) the two launches finish sooner (``tools/dropin_latency.py``).  None (default): two
    launches for a warm-started CHAIN, one for independent cold starts - ``warm_start=False``, and the cold starts a
    warm-started sweep is first solved as (``SolverConfig.parallel_chains``, ``_segment_length``).
    """
    import torch

    cfg = _coerce_config(solver_config)
    with_extra = (lambda states, infos, extra=None: (states, infos, extra)) if evaluation is not None else (lambda states, infos, extra=None: (states, infos))
    if sweep_config.n_steps == 0:
        return with_extra([], [])
    segments = [(0, sweep_config.n_steps)] if _flattened is not None else target_segments(sweep_config)  # (a memo hit is a one-run sweep)
    if len(segments) > 1:
        return with_extra(*_solve_sweep_in_runs(initial_state, constraints, sweep_config, derived_manager, cfg, segments,
                                                output_points, device))
    # (`_flattened`: the program and target table sweep.solve_sweep found in its per-suspension memo - `constraints` is then unused)
    program, table = _flattened if _flattened is not None else dropin_program(initial_state, constraints, sweep_config, derived_manager,
                                                                             cfg, output_points)
    dp = _device_program(program, device)
    n_steps = table.shape[0]
    solve, evaluated, fused = dp.solve, None, False
    one_launch = (not cfg.warm_start) if evaluation_fused is None else bool(evaluation_fused)
    if evaluation is not None and program.n_targets > 0:
        try:
            roles, want_tangents = evaluation(program)
            dp.enable_evaluation(roles)

            def solve(targets, one_launch=one_launch, **kw):  # noqa: F811 - the same launch, ending in the evaluation epilogue
                if one_launch:
                    return dp.solve_evaluated(targets, tangents=want_tangents, **kw)
                plain = dp.solve(targets, **kw)   # (both launches are asynchronous: the records never leave HBM in between)
                return dp.evaluate(plain.positions, tangents=want_tangents, info_raw=plain.info_raw)

            fused = True
        except (ValueError, RuntimeError):  # no evaluated kernels for this program: solve now, evaluate after
            solve, fused = dp.solve, False
    solve_kw = dict(max_iter=cfg.max_iter, residual_tolerance=cfg.residual_tolerance, predictor=False,
                    **device_tolerances(cfg, program))  # (one sweep = a few chains or explicit cold starts: nothing for a fitted model to save)
    targets = torch.as_tensor(table)
    positions = info = None
    positions_from_segments = False
    segment = _segment_length(n_steps) if cfg.warm_start and cfg.parallel_chains else 0
    if segment:
        # One chain is one quad walking the sweep step by step: 1/16384 of the chip and ~12 us per step.  The sweep
        # is therefore solved as chains of `segment` steps side by side (`_segment_length`: one step - every state a
        # cold start from the design state) and the result is kept only if it is what the sequential warm start would
        # have produced: every step accepted and every head where the secant through its neighbours says it should be.
        # Single-step chains are independent cold starts: their evaluation rides in the solve's launch unless the
        # caller chose a form.
        attempt_kw = {"one_launch": True} if fused and evaluation_fused is None and segment == 1 else {}
        result = solve(targets, chain_len=segment, **attempt_kw, **solve_kw)
        positions, info = result.host()  # (one wait for both copies)
        positions_from_segments = True
        if not _chains_are_continuous(program, table, positions, info, segment):
            positions = info = None
            positions_from_segments = False
    if positions is None:
        result = solve(targets, chain=bool(cfg.warm_start), **solve_kw)
        positions, info = result.host()
        if not cfg.warm_start:
            positions_from_segments, segment = True, 1  # every step is a chain head
    _raise_on_first_failure(program, dp, table, positions, info, sweep_config, initial_state, cfg)
    states = _states_from_positions(initial_state, program, positions)
    if fused:  # (never `solve is not dp.solve`: a bound method is a new object on every attribute access)
        evaluated = (program, result)
    return with_extra(states, _solver_infos(info, dp, segment if positions_from_segments else n_steps), evaluated)


def _solver_infos(info: np.ndarray, dp, chain_len: int) -> list:
    """``SolverInfo`` per step.  A chain head that took its first step from the program's shared first-step table
    (``okx_solve_opts.shared_first_step``) did not run the design-state evaluation itself - the device's ``nfev`` counts
    what a problem ran - but the evaluation was made on its behalf: it is counted here, like the reference counts it."""
    nfev = info["nfev"].astype(np.int64)
    if getattr(dp, "shares_first_step", False) and chain_len > 0:
        nfev[::chain_len] += 1
    return [SolverInfo(c, n, r) for c, n, r in zip(((info["flags"] & 1) != 0).tolist(), nfev.tolist(),
                                                   info["max_residual"].tolist())]


def _solve_sweep_in_runs(initial_state, constraints, sweep_config, derived_manager, cfg, segments, output_points, device):
    """
    A sweep whose target rows change between steps (``targeting.py:67-75`` pairs arbitrary targets by index): every
    maximal run of steps with one set of target rows is one warm-started chain on its own program, and the run after it
    starts where this one ended, exactly like the reference's sequential warm start (``solver.py:716,774``) - the next
    program is flattened at the last solved state (its constraint rows keep their authored parameters; only the start
    point moves).  Relative targets stay relative to the ORIGINAL initial state (``solver.py:584-627``).
    """
    import torch

    spec = getattr(derived_manager, "spec", derived_manager)
    n_vars = 3 * len(initial_state.free_points)
    states, infos = [], []
    start_state = initial_state
    keep = None if output_points is None else set(output_points)
    for lo, hi in segments:
        heads, table = absolute_target_table(sweep_config, initial_state, (lo, hi))
        validate_least_squares_dimensions(n_vars, len(constraints) + len(heads))
        program = flatten_problem(start_state, constraints, spec, heads, output_points=None,
                                  line_mode="softnorm").with_line_mode(cfg.line_mode)
        dp = _device_program(program, device)
        result = dp.solve(torch.as_tensor(table), chain=bool(cfg.warm_start), max_iter=cfg.max_iter,
                          residual_tolerance=cfg.residual_tolerance, predictor=False, **device_tolerances(cfg, program))
        positions = result.positions.cpu().numpy()
        info = result.info()
        _raise_on_first_failure(program, dp, table, positions, info, sweep_config, initial_state, cfg, first_step=lo)
        run_states = _states_from_positions(initial_state, program, positions)
        start_state = run_states[-1]
        infos.extend(_solver_infos(info, dp, hi - lo if cfg.warm_start else 1))
        states.extend(run_states)
    if keep is not None:
        states = [type(s)(positions={k: v for k, v in s.positions.items() if k in keep},
                          free_points=set(s.free_points) & keep) for s in states]
    return states, infos


def _segment_length(n_steps: int) -> int:
    """
    Chain length for a warm-started sweep solved as several chains at once (0: keep it one chain).  One chain step is
    ~12 us of ONE wavefront (~40 us in a composed axle's pair mode) whatever else the chip does, so a sweep's time on
    the device is the length of its longest chain - and a sweep the drop-in sees (tens to hundreds of steps) leaves the
    chip empty either way.  Every step therefore starts cold from the design state (chains of ONE: the cold body, the
    shared first step) and the continuity test decides whether that is the path the sequential warm start walks:
    101-step corner sweep 0.47 ms per call as ten chains of eleven, 0.38 as chains of four, 0.34 as cold starts; a
    31-step T-bar axle 1.39 (one chain) / 0.47 / 0.33 ms (`tools/dropin_phases.py --segment=k`,
    `tools/dropin_latency.py --segment=k`).  Below four steps the test has nothing to hold the second state against.
    """
    return 1 if n_steps >= 4 else 0


def _chains_are_continuous(program: ConstraintProgram, table: np.ndarray, positions: np.ndarray, info: np.ndarray,
                           segment: int) -> bool:
    """
    Whether a sweep solved as chains of `segment` steps is the path the sequential warm start (reference
    ``solver.py:716,774``) follows: every step accepted, and at every chain boundary the head's free coordinates within
    half a step of the secant extrapolation of the previous chain's last two states (a head that fell onto another
    assembly branch is many steps away from it).
    """
    if not np.all((info["flags"] & 7) == 1):  # converged, neither residual-exceeded nor failed - every step
        return False
    n = positions.shape[0]
    if segment >= n:
        return True
    rows = program.__dict__.get("_okx_free_rows", False)  # (where the free points sit among the outputs: kept on the program)
    if rows is False:
        out = [int(k) for k in program.out_point]
        try:
            rows = np.asarray([out.index(int(p)) for p in program.free_point], dtype=np.intp)
            if rows.size and np.array_equal(rows, np.arange(rows[0], rows[0] + rows.size)):
                rows = slice(int(rows[0]), int(rows[0]) + rows.size)  # (a view instead of a gather)
        except ValueError:  # a free point is not among the outputs: nothing to check against
            rows = None
        program.__dict__["_okx_free_rows"] = rows
    if rows is None:
        return False
    free = positions[:, rows, :].reshape(n, -1)

    def deviates(head, last, before, t_head, t_last, t_before) -> bool:
        """Any head further than half a step from the secant through `before` and `last` (rows of states / target rows)."""
        d_prev, d_new = t_last - t_before, t_head - t_last
        den = (d_prev * d_prev).sum(axis=1)
        alpha = (d_new * d_prev).sum(axis=1) / np.where(den > 0.0, den, np.inf)
        step = last - before
        scale = np.abs(step).max(axis=1) * np.maximum(np.abs(alpha), 1.0)
        return bool(np.any(np.abs(head - last - alpha[:, None] * step).max(axis=1) > 0.5 * scale + 1e-6))

    # a head is held against the secant through the two states BEFORE it; the first two states of a sweep of single-step
    # chains are held against the two states AFTER them (the same test, mirrored).
    # Every boundary at once (a loop over the heads cost as much host time as the launch).
    if segment == 1:  # every state a head: consecutive rows, no gathers
        if n < 4:
            return False
        # (the first state too: it is the sequential path's own cold start, and holding it against the two states after
        #  it is what ties the rest to its branch when the first target increment is much larger than the second)
        return not (deviates(free[2:], free[1:-1], free[:-2], table[2:], table[1:-1], table[:-2])
                    or deviates(free[0:2], free[1:3], free[2:4], table[0:2], table[1:3], table[2:4]))
    heads = np.arange(segment, n, segment)
    return not deviates(free[heads], free[heads - 1], free[heads - 2], table[heads], table[heads - 1], table[heads - 2])


def _coerce_config(config) -> SolverConfig:
    if isinstance(config, SolverConfig):
        return config
    fields = {k: getattr(config, k) for k in SolverConfig._fields if hasattr(config, k)}
    return SolverConfig(**fields)


def _raise_on_first_failure(program, dp, table, positions, info, sweep_config, initial_state, cfg, first_step: int = 0) -> None:
    import torch

    flags = info["flags"]
    bad = np.nonzero(((flags & 1) == 0) | ((flags & 6) != 0))[0]
    if bad.size == 0:
        return
    step = int(bad[0])
    step_targets = convert_targets_to_absolute([dim[first_step + step] for dim in sweep_config.target_sweeps], initial_state)
    # A step that stalls far from feasibility is the reference's "converged to a compromise"
    # case (MINPACK stops on ftol=1e-5 there): report it through the residual check.
    exceeded = (flags[step] & 2) != 0 and (flags[step] & 4) == 0
    if not exceeded:
        raise RuntimeError(
            f"Solver failed to converge for targets: {step_targets}."
            f"\nMessage: device Levenberg-Marquardt stopped after {int(info['iterations'][step])} iterations"
        )
    # the record holds the OUTPUT points in output order: map the free points through it (a restricted output list may
    # not hold all of them; the worst row is then named from a re-solve of this one step with every point written)
    out = [int(k) for k in program.out_point]
    try:
        x = positions[step][[out.index(int(p)) for p in program.free_point]].reshape(1, -1)
    except ValueError:
        x = None
    if x is not None:
        r, _ = dp.eval(torch.as_tensor(x), torch.as_tensor(table[step : step + 1]), jac=False)
        worst = describe_worst_residual(program, r.cpu().numpy()[0])
    else:
        worst = "unavailable (the requested output_points do not include every free point)"
    value = float(info["max_residual"][step])
    step = first_step + step
    raise RuntimeError(
        f"Solve at sweep step {step} did not reach an acceptable residual: worst residual "
        f"{value:.6g} exceeds the acceptance tolerance {cfg.residual_tolerance:.6g}. "
        f"Worst residual row: {worst}. The mechanism likely cannot reach the requested targets "
        "(kinematic lock-out / infeasible target combination)."
    )


def _states_from_positions(initial_state, program: ConstraintProgram, positions: np.ndarray) -> list:
    """One independent state per step, typed like the input state (``solver.py:763``).  The points of a state are made
    when they are first read (``state.RowPositions``): a 101-step sweep hands back 101 small objects, not 2000."""
    from .state import RowPositions

    sample = next(iter(initial_state.positions.values()))
    point_cls = type(sample) if hasattr(sample, "data") else Point3
    own_state = not hasattr(initial_state, "free_points_order") or type(initial_state) is SuspensionState
    state_cls = SuspensionState if own_state else type(initial_state)
    keys = [program.point_keys[k] for k in program.out_point]
    index = {k: i for i, k in enumerate(keys)}
    free = set(initial_state.free_points) & set(keys)
    order = sorted(free)
    # one contiguous copy of the whole sweep; every point of every state is its own view into it
    # (independent of the device buffer and of every other state, like the reference's per-step copies)
    block = np.array(positions, dtype=np.float64, order="C", copy=True)
    make = getattr(point_cls, "from_trusted", point_cls)  # this package's Point3: adopt the view, no second copy
    states = []
    for step in range(block.shape[0]):
        lazy = RowPositions(block[step], index, make)
        if own_state:  # (what the dataclass's __init__ / __post_init__ do, without sorting the same keys once per step)
            state = SuspensionState.__new__(SuspensionState)
            state.positions, state.free_points, state.free_points_order = lazy, set(free), list(order)
        else:
            state = state_cls(positions=lazy, free_points=set(free))
        states.append(state)
    return states
