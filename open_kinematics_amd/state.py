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


class RowPositions(dict):
    """
    ``dict[point key -> Point3]`` over one state's block ``rows [n_points, 3]`` of a solved sweep's host array: a ``Point3``
    is made (a VIEW of its row - no copy, as before) the first time its key is read and kept from then on, so a sweep's
    states cost one small object each instead of one per point until somebody looks at the points.  It behaves as the plain
    dict the reference's ``SuspensionState.positions`` is - lookup, ``in``, ``len``, iteration in point order, ``items()``,
    ``dict(...)``, ``==``, assignment, deletion, ``copy()`` - and a write replaces the entry like in any dict: the block
    underneath belongs to this sweep's result alone (``solver.py:763``: independent copies per step).
    """

    __slots__ = ("_rows", "_index", "_make", "_gone")

    def __init__(self, rows: np.ndarray, index: dict, make) -> None:
        super().__init__()
        self._rows, self._index, self._make, self._gone = rows, index, make, None

    def __missing__(self, key):
        i = self._index.get(key)
        if i is None or (self._gone is not None and key in self._gone):
            raise KeyError(key)
        point = self._make(self._rows[i])
        dict.__setitem__(self, key, point)
        return point

    def _lazy(self, key) -> bool:
        return key in self._index and not (self._gone is not None and key in self._gone)

    def __contains__(self, key) -> bool:
        return dict.__contains__(self, key) or self._lazy(key)

    def __iter__(self):
        for key in self._index:
            if dict.__contains__(self, key) or self._lazy(key):
                yield key
        for key in dict.__iter__(self):  # keys assigned by the caller that the sweep did not write
            if key not in self._index:
                yield key

    def __len__(self) -> int:
        extra = sum(1 for key in dict.__iter__(self) if key not in self._index or (self._gone is not None and key in self._gone))
        return len(self._index) - (len(self._gone) if self._gone else 0) + extra

    def __setitem__(self, key, value) -> None:
        dict.__setitem__(self, key, value)

    def __delitem__(self, key) -> None:
        if dict.__contains__(self, key):
            dict.__delitem__(self, key)
        elif not self._lazy(key):
            raise KeyError(key)
        if key in self._index:
            if self._gone is None:
                self._gone = set()
            self._gone.add(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        from collections.abc import KeysView

        return KeysView(self)

    def items(self):
        from collections.abc import ItemsView

        return ItemsView(self)

    def values(self):
        from collections.abc import ValuesView

        return ValuesView(self)

    def pop(self, key, *default):
        try:
            value = self[key]
        except KeyError:
            if default:
                return default[0]
            raise
        del self[key]
        return value

    def popitem(self):
        for key in reversed(list(self)):
            return key, self.pop(key)
        raise KeyError("popitem(): dictionary is empty")

    def setdefault(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            self[key] = default
            return default

    def update(self, *args, **kwargs) -> None:
        for key, value in dict(*args, **kwargs).items():
            self[key] = value

    def clear(self) -> None:
        dict.clear(self)
        self._gone = set(self._index)

    def copy(self) -> dict:
        return dict(self.items())

    def __eq__(self, other) -> bool:
        if not isinstance(other, dict):
            return NotImplemented
        return dict(self.items()) == (dict(other.items()) if isinstance(other, RowPositions) else other)

    def __ne__(self, other) -> bool:
        result = self.__eq__(other)
        return result if result is NotImplemented else not result

    __hash__ = None

    def __repr__(self) -> str:
        return repr(dict(self.items()))

    def __reduce__(self):
        return (dict, (dict(self.items()),))

    def rows_if_untouched(self):
        """``(rows, index)`` while no point of this state has been read, assigned or deleted yet - what a caller that wants
        every point as ONE array takes instead of walking the points (``sensitivity._positions_array``); else None."""
        if self._gone or dict.__len__(self):
            return None
        return self._rows, self._index


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
