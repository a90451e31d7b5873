"""``SuspensionState`` and a minimal ``Point3`` (reference ``core/state.py``,
``core/primitives/geometry.py``): the container types the sweep API hands back."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Set

import numpy as np


class Point3:
    """A position in 3-D space; ``.data`` is a float64[3] array."""

    __slots__ = ("data",)

    def __init__(self, data) -> None:
        raw = np.array(getattr(data, "data", data), dtype=np.float64)
        if raw.shape != (3,):
            raise ValueError(f"Point3 requires shape (3,), got {raw.shape}")
        self.data = raw

    @classmethod
    def from_trusted(cls, data: np.ndarray) -> "Point3":
        obj = cls.__new__(cls)
        obj.data = data
        return obj

    def copy(self) -> "Point3":
        return Point3.from_trusted(self.data.copy())

    def __getitem__(self, idx) -> float:
        return float(self.data[int(idx)])

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def __sub__(self, other) -> np.ndarray:
        return self.data - np.asarray(getattr(other, "data", other))

    @property
    def x(self) -> float:
        return float(self.data[0])

    @property
    def y(self) -> float:
        return float(self.data[1])

    @property
    def z(self) -> float:
        return float(self.data[2])

    def __repr__(self) -> str:
        return f"Point3({self.data})"


@dataclass
class SuspensionState:
    """All point positions of one solved (or design) state (``core/state.py:23-72``)."""

    positions: dict[Any, Point3]
    free_points: Set[Any]
    free_points_order: List[Any] = field(init=False)

    def __post_init__(self) -> None:
        self.free_points_order = sorted(list(self.free_points))

    @property
    def fixed_points(self) -> Set[Any]:
        return set(self.positions.keys()) - self.free_points

    def get_free_array(self) -> np.ndarray:
        return np.concatenate([self.positions[k].data for k in self.free_points_order])

    def update_from_array(self, array: np.ndarray) -> None:
        n = len(self.free_points_order)
        if array.shape != (n * 3,):
            raise ValueError(f"Array shape {array.shape} doesn't match expected ({n * 3},)")
        rows = array.reshape(n, 3)
        for i, key in enumerate(self.free_points_order):
            self.positions[key] = Point3.from_trusted(rows[i])

    def copy(self) -> "SuspensionState":
        return SuspensionState(
            positions={k: v.copy() for k, v in self.positions.items()},
            free_points=set(self.free_points),
        )

    def get(self, key) -> Point3:
        return self.positions[key]

    def set(self, key, position) -> None:
        self.positions[key] = Point3(position)

    __getitem__ = get

    def __setitem__(self, key, position) -> None:
        self.set(key, position)

    def __contains__(self, key) -> bool:
        return key in self.positions

    def items(self):
        return self.positions.items()

    def keys(self):
        return self.positions.keys()

    def values(self):
        return self.positions.values()
