"""Identifiers shared by the host-side API (names and values follow reference ``core/enums.py``
and ``core/primitives/point_ref.py`` so that output column names and key ordering match)."""

from __future__ import annotations

from enum import Enum, IntEnum
from typing import NamedTuple, Union


class Axis(IntEnum):
    X = 0
    Y = 1
    Z = 2


class TargetPositionMode(str, Enum):
    RELATIVE = "relative"
    ABSOLUTE = "absolute"

    def __str__(self) -> str:
        return self.value


class PointID(IntEnum):
    """Authored and derived suspension points (reference ``core/enums.py:33-81``)."""

    NOT_ASSIGNED = 0
    LOWER_WISHBONE_INBOARD_FRONT = 1
    LOWER_WISHBONE_INBOARD_REAR = 2
    LOWER_WISHBONE_OUTBOARD = 3
    UPPER_WISHBONE_INBOARD_FRONT = 4
    UPPER_WISHBONE_INBOARD_REAR = 5
    UPPER_WISHBONE_OUTBOARD = 6
    PUSHROD_INBOARD = 7
    PUSHROD_OUTBOARD = 8
    TRACKROD_INBOARD = 9
    TRACKROD_OUTBOARD = 10
    TOE_LINK_INBOARD = 11
    TOE_LINK_OUTBOARD = 12
    AXLE_INBOARD = 13
    AXLE_OUTBOARD = 14
    AXLE_MIDPOINT = 15
    STRUT_TOP = 16
    STRUT_BOTTOM = 17
    WHEEL_CENTER = 18
    WHEEL_INBOARD = 19
    WHEEL_OUTBOARD = 20
    CONTACT_PATCH_CENTER = 21
    CAMBER_SHIM_FACE_POINT_A = 22
    CAMBER_SHIM_FACE_POINT_B = 23
    CAMBER_SHIM_FACE_NORMAL = 24
    ROCKER_AXIS_A = 25
    ROCKER_AXIS_B = 26
    DROPLINK_ROCKER = 27
    DROPLINK_U_BAR = 28
    ARB_U_BAR_AXIS_A = 29
    ARB_U_BAR_AXIS_B = 30
    HEAVE_LINK_ROCKER = 31
    ARB_T_BAR_PIVOT = 32
    DROPLINK_T_BAR = 33


class Side(IntEnum):
    """ISO 8855: LEFT is +Y.  Ordered LEFT < RIGHT < CENTER (``point_ref.py:24-52``)."""

    LEFT = 0
    RIGHT = 1
    CENTER = 2


class PointRef(NamedTuple):
    """Side-qualified point key of an axle model (``point_ref.py:55-86``)."""

    side: Side
    point: PointID

    @property
    def name(self) -> str:
        return f"{self.side.name}_{self.point.name}"


PointKey = Union[PointID, PointRef]


def point_key_name(key) -> str:
    return key.name.lower()
