"""
Drop-in for ``kinematics.core.sensitivity`` (reference ``core/sensitivity.py:27-143``) and
``kinematics.core.sweep.compute_sweep_tangents`` (``core/sweep.py:113-141``): solution-manifold
tangents of solved sweep states, computed for the whole sweep by ONE launch of the generated
tangent kernel (``okx_tangent_batch``; ``csrc/okx_quadgen.cpp``).

Same math as the reference: with ``J`` the analytical residual Jacobian at a solved state,
``J q_t = e_t`` in the least-squares sense for every target row ``t``; the pinned line rows of the
device program play the role of ``_degenerate_constraint_pins`` (``sensitivity.py:146-174``) — they
span the same plane with the same Gram matrix, so ``J^T J`` and the velocities are identical
(``tests/test_tangents_oracle.py``).  Derived-point velocities come from closed-form forward-mode
derivatives instead of the dual-number pass.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np


@dataclass(frozen=True)
class TangentField:
    """First-order response of every point position to one sweep target (``sensitivity.py:27-40``)."""

    target_index: int
    target: Any
    velocities: dict

    def velocity(self, point_id) -> np.ndarray:
        velocity = self.velocities.get(point_id)
        if velocity is None:
            return np.zeros(3, dtype=np.float64)
        return velocity


@dataclass(frozen=True)
class TangentSolveInfo:
    """
    Numerical health of one state's tangent solve (``sensitivity.py:43-55``).  The device factors
    ``J^T J`` (LDL^T) instead of taking an SVD of ``J``: its pivots lie inside ``[s_min^2, s_max^2]``,
    so ``smallest_singular_value`` is reported as ``sqrt(min_pivot)`` (never below ``s_min``) and
    ``condition_number`` as ``sqrt(max_pivot / min_pivot)`` (never above ``cond(J)``); ``rank`` is
    ``n_variables`` unless a pivot vanished.
    """

    n_variables: int
    rank: int
    smallest_singular_value: float
    condition_number: float

    @property
    def rank_deficient(self) -> bool:
        return self.rank < self.n_variables


@dataclass(frozen=True)
class SweepTangents:
    """``core/sweep.py:68-76``."""

    per_step: list
    solve_infos: list


def solve_infos_from_records(info: np.ndarray, n_variables: int) -> list:
    """
    Device tangent records (``okx_tangent_info``: pivot range of the LDL^T of ``J^T J`` + flags) as the
    reference's per-state ``TangentSolveInfo`` (``sensitivity.py:44-55``).  The pivots lie inside
    ``[s_min^2, s_max^2]`` of ``J``, so their square roots stand in for the singular values; a rank-deficient
    factorisation reports ``rank = n - 1`` (the device does not count how many pivots vanished).
    """
    from ._abi import TANGENT_RANK_DEFICIENT

    lo, hi = info["min_pivot"].astype(np.float64), info["max_pivot"].astype(np.float64)
    positive = lo > 0.0
    smallest = np.sqrt(np.maximum(lo, 0.0)).tolist()
    with np.errstate(divide="ignore", invalid="ignore"):
        condition = np.where(positive, np.sqrt(hi / np.where(positive, lo, 1.0)), np.inf).tolist()
    deficient = ((info["flags"] & TANGENT_RANK_DEFICIENT) != 0).tolist()
    return [TangentSolveInfo(n_variables=n_variables, rank=n_variables - 1 if d else n_variables,
                             smallest_singular_value=s, condition_number=c)
            for d, s, c in zip(deficient, smallest, condition)]


def _positions_array(states, out_keys) -> np.ndarray:
    rows = []
    # states straight from solve_sweep whose points nobody has touched: their blocks ARE the array (state.RowPositions)
    picks, cached_index = [], None
    for state in states:
        untouched = getattr(state.positions, "rows_if_untouched", None)
        block = untouched() if untouched is not None else None
        if block is None:
            picks = None
            break
        if block[1] is not cached_index:
            cached_index, take = block[1], [block[1].get(k) for k in out_keys]
            if any(i is None for i in take):
                picks = None
                break
        picks.append(block[0] if take == list(range(block[0].shape[0])) else block[0][take])
    if picks is not None and (picks or not states):
        return np.asarray(picks, dtype=np.float64).reshape(len(states), len(out_keys), 3)
    for state in states:
        rows.append([np.asarray(getattr(state.positions[k], "data", state.positions[k]), dtype=np.float64)
                     for k in out_keys])
    return np.asarray(rows, dtype=np.float64).reshape(len(states), len(out_keys), 3)


def compute_sweep_tangents(suspension, sweep_config, states, *, device=None) -> SweepTangents:
    """
    Tangent fields of every solved state of a sweep (``core/sweep.py:113-141``): one
    ``TangentField`` per target and step, velocities keyed like ``state.positions``.
    """
    import torch

    from .solver import _device_program, convert_targets_to_absolute
    from .sweep import sweep_program

    program, _ = sweep_program(suspension, sweep_config)
    dp = _device_program(program, device)
    out_keys = [program.point_keys[k] for k in program.out_point]
    pos = _positions_array(states, out_keys)
    tan, tinfo = dp.tangents(torch.as_tensor(pos, device=dp.device))
    tan = tan.cpu().numpy()
    infos = solve_infos_from_records(dp.tangent_info(tinfo), program.n_vars)
    initial = suspension.initial_state()
    per_step = []
    for s in range(len(states)):
        step_targets = convert_targets_to_absolute([sweep[s] for sweep in sweep_config.target_sweeps], initial)
        fields = []
        for t, target in enumerate(step_targets):
            velocities = {key: tan[s, t, k].copy() for k, key in enumerate(out_keys)}
            fields.append(TangentField(target_index=t, target=target, velocities=velocities))
        per_step.append(fields)
    return SweepTangents(per_step=per_step, solve_infos=infos)


def compute_state_tangents(state, constraints, derived_manager, step_targets: Sequence[Any], *, device=None):
    """
    ``sensitivity.py:57-143`` for one state: ``(list[TangentField], TangentSolveInfo)``.
    ``derived_manager`` may be the reference's ``DerivedPointsManager`` or a derived spec.
    """
    import torch

    from .batch import DeviceProgram
    from .program import flatten_problem

    if not step_targets:
        return [], TangentSolveInfo(n_variables=0, rank=0, smallest_singular_value=0.0, condition_number=1.0)
    heads = [(t.point_id, t.direction) for t in step_targets]
    out_keys = list(state.positions.keys())
    program = flatten_problem(state, constraints, derived_manager, heads, out_keys,
                              line_mode="softnorm").with_line_mode("pinned")
    # The state being differentiated is this program's "design" state, so every call is its own program:
    # created and released here, not through the drop-in cache (it would only evict the sweep programs).
    dp = DeviceProgram(program, device)
    try:
        keys = [program.point_keys[k] for k in program.out_point]
        pos = _positions_array([state], keys)
        tan, tinfo = dp.tangents(torch.as_tensor(pos, device=dp.device))
        tan = tan.cpu().numpy()[0]
        info = solve_infos_from_records(dp.tangent_info(tinfo), program.n_vars)[0]
    finally:
        dp.close()
    fields = [TangentField(target_index=t, target=target,
                           velocities={key: tan[t, k].copy() for k, key in enumerate(keys)})
              for t, target in enumerate(step_targets)]
    return fields, info


def combine_tangents(fields: Sequence[TangentField], coefficients: Sequence[float]) -> dict:
    """
    The velocity field ``sum_k coefficients[k] * fields[k]`` as ``{point key: velocity}`` (the reference's
    ``combine_tangents``, ``sensitivity.py:177-196``).  The fields of one state hold the rows ``[T, n_points, 3]`` of one
    device tensor: they are stacked back into that array and contracted in one product; a point that only some fields
    carry counts as zero in the others.
    """
    weights = np.asarray(list(coefficients), dtype=np.float64)
    if len(fields) != weights.shape[0]:
        raise ValueError(f"Field/coefficient count mismatch: {len(fields)} fields, {weights.shape[0]} coefficients.")
    keys: list = []
    seen = set()
    for field in fields:
        for key in field.velocities:
            if key not in seen:
                seen.add(key)
                keys.append(key)
    stacked = np.zeros((len(fields), len(keys), 3), dtype=np.float64)
    column = {key: k for k, key in enumerate(keys)}
    for row, field in zip(stacked, fields):
        for key, velocity in field.velocities.items():
            row[column[key]] = velocity
    combined = np.tensordot(weights, stacked, axes=(0, 0))
    return {key: combined[k] for k, key in enumerate(keys)}
