"""Sweep targets (reference ``core/targeting.py``)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, NamedTuple, Union

import numpy as np

from .enums import Axis, TargetPositionMode
from .program import resolve_direction

EPS_GEOMETRIC = 1e-6


@dataclass(frozen=True)
class PointTargetAxis:
    axis: Axis


@dataclass(frozen=True)
class PointTargetVector:
    vector: Any  # unit 3-vector (array or object with .data)


PointTargetDirection = Union[PointTargetAxis, PointTargetVector]


class PointTarget(NamedTuple):
    """One scalar target: ``dot(position(point_id), direction) == value`` (``targeting.py:86-104``)."""

    point_id: Any
    direction: PointTargetDirection
    value: float
    mode: TargetPositionMode = TargetPositionMode.RELATIVE


@dataclass
class SweepConfig:
    """``target_sweeps[dimension][step]``; all dimensions index-paired (``targeting.py:51-84``)."""

    target_sweeps: list

    def __post_init__(self):
        if not self.target_sweeps:
            return
        lengths = [len(s) for s in self.target_sweeps]
        if len(set(lengths)) > 1:
            raise ValueError(f"All sweep dimensions must have the same length. Got: {lengths}")

    @property
    def n_steps(self) -> int:
        return len(self.target_sweeps[0]) if self.target_sweeps else 0


def resolve_target(direction) -> np.ndarray:
    """Unit world direction of a target specification (``targeting.py:135-148``)."""
    return resolve_direction(direction)


@dataclass(frozen=True)
class ActuatorDOF:
    """One physical actuator coordinate a sweep must control (``targeting.py:151-166``)."""

    name: str
    point_keys: tuple
    direction: Any

    def matches(self, target) -> bool:
        if target.point_id not in self.point_keys:
            return False
        d = np.asarray(getattr(self.direction, "data", self.direction), dtype=np.float64)
        return abs(float(np.dot(resolve_direction(target.direction), d))) >= 1.0 - EPS_GEOMETRIC


def validate_sweep_controls(sweep_config, actuator_dofs) -> None:
    """Exactly one target per physical actuator coordinate and step (``targeting.py:168-186``)."""
    for actuator in actuator_dofs:
        memo: dict = {}  # (point, direction object) -> matches: a dimension repeats both for every step
        for step in range(sweep_config.n_steps):
            hits = 0
            for dim in sweep_config.target_sweeps:
                target = dim[step]
                key = (target.point_id, id(target.direction))
                hit = memo.get(key)
                if hit is None:
                    hit = memo[key] = _matches(actuator, target)
                hits += hit
            if hits != 1:
                raise ValueError(
                    f"Sweep requires exactly one target for actuator '{actuator.name}' along "
                    f"its motion axis; found {hits} at step {step}."
                )


def _matches(actuator, target) -> bool:
    matcher = getattr(actuator, "matches", None)
    if matcher is not None and not isinstance(actuator, ActuatorDOF):
        return bool(matcher(target))  # a reference ActuatorDOF with reference targets
    return ActuatorDOF.matches(actuator, target)
