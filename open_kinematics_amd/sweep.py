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
    return solve_suspension_sweep(
        initial_state=suspension.initial_state(),
        constraints=suspension.constraints(),
        sweep_config=sweep_config,
        derived_manager=suspension.derived_spec(),
        solver_config=solver_config,
        device=device,
    )


def sweep_program(suspension, sweep_config, line_mode: str = "pinned") -> tuple[ConstraintProgram, np.ndarray]:
    """Constraint program (outputs = ``suspension.output_points()``) and absolute targets ``[S, T]``."""
    validate_sweep_controls(sweep_config, suspension.actuator_dofs())
    state = suspension.initial_state()
    heads, table = absolute_target_table(sweep_config, state)
    program = flatten_problem(state, suspension.constraints(), suspension.derived_spec(), heads,
                              suspension.output_points(), line_mode="softnorm").with_line_mode(line_mode)
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
