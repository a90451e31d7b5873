"""
Problem emission for the suspension topologies of the BASELINE configs: which points are
free, which constraint rows exist (in the reference's emission order, so that row indices
line up with the reference's Jacobian rows), and which derived points are declared.

Reference (``core/suspensions/``): ``corner/double_wishbone.py:233-308``,
``corner/macpherson.py:247-323``, ``corner/track_rod.py:60-97``, ``corner/toe_link.py:58-86``,
``corner/attachments.py:23-121``, ``corner/mechanisms.py:267-302``,
``axle/suspension.py:146-211,253-268``, ``axle/mechanisms.py:281-342``, ``build.py:297-354``.

Scope: double-wishbone corner (direct or pushrod-rocker actuation; no spring, coil-over or
torsion bar; rack or fixed toe link), MacPherson corner, and the composed axle with rack
coupling and a U-bar anti-roll bar; a double wishbone's camber-shim setup solve runs on the device
(``shims.py``); the axle's shared hardware: U-bar or rigid T-bar anti-roll bar, rocker-to-rocker heave link.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from functools import partial
from typing import Any, Sequence

import numpy as np

from .constraints import (
    AngleConstraint,
    Constraint,
    DistanceConstraint,
    MidpointOnPlaneConstraint,
    PointOnLineConstraint,
    ScalarTripleProductConstraint,
)
from .derived import (
    DerivedPointsManager,
    DerivedPointsSpec,
    build_wheel_derived_spec,
    get_point_along_line,
)
from .enums import PointID, PointRef, Side
from .state import Point3, SuspensionState
from .targeting import ActuatorDOF

P = PointID
EPS_GEOMETRIC = 1e-6
MIN_CHIRALITY_VOLUME = 1e-6
MM_PER_INCH = 25.4
WORLD_Y = np.array([0.0, 1.0, 0.0])


def _d(positions, a, b) -> float:
    """Design length between two points (``geometric.py:17-28``)."""
    return float(np.linalg.norm(positions[b].data - positions[a].data))


def _distance(positions, a, b) -> DistanceConstraint:
    return DistanceConstraint(a, b, _d(positions, a, b))


def _angle(v1: np.ndarray, v2: np.ndarray) -> float:
    """``compute_vector_vector_angle`` (``geometric.py:71-104``)."""
    u1 = v1 / float(np.linalg.norm(v1))
    u2 = v2 / float(np.linalg.norm(v2))
    return float(np.arctan2(float(np.linalg.norm(np.cross(u1, u2))), float(np.dot(u1, u2))))


def _triple(v1, v2, v3) -> float:
    return float(np.dot(v1, np.cross(v2, v3)))


def chiral_rigid_point_constraints(positions, point, references) -> list[Constraint]:
    """Three design distances plus the authored handedness (``attachments.py:23-74``)."""
    rows: list[Constraint] = [_distance(positions, point, ref) for ref in references]
    a, b, c = (positions[r].data for r in references)
    volume = _triple(b - a, c - a, positions[point].data - a)
    if abs(volume) < MIN_CHIRALITY_VOLUME:
        raise ValueError(f"{point.name} and its rigid-body references do not define reliable handedness")
    rows.append(ScalarTripleProductConstraint(*references, point, target_volume=volume, scale=abs(volume)))
    return rows


def anchored_rigid_point_constraints(positions, point, anchors) -> list[Constraint]:
    """First three anchors chiral, further anchors plain distances (``attachments.py:97-121``)."""
    rows = chiral_rigid_point_constraints(positions, point, tuple(anchors[:3]))
    rows.extend(_distance(positions, point, anchor) for anchor in anchors[3:])
    return rows


def validate_rigid_anchor_points(hardpoints, anchors, label: str) -> None:
    """``attachments.py:77-94``."""
    if len(anchors) < 3:
        raise ValueError(f"{label} requires at least three mounting body anchors")
    a, b, c = (hardpoints[p].data for p in anchors[:3])
    if float(np.linalg.norm(b - a)) <= EPS_GEOMETRIC:
        raise ValueError(f"{label} mounting body anchors must be distinct")
    line = (b - a) / float(np.linalg.norm(b - a))
    if float(np.linalg.norm(np.cross(c - a, line))) <= EPS_GEOMETRIC:
        raise ValueError(f"The first three {label} mounting body anchors must not be collinear")


# --------------------------------------------------------------------------------------
# corner mechanisms
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class HeadingLink:
    """Track rod (rack-driven) or fixed toe link (``track_rod.py`` / ``toe_link.py``)."""

    steered: bool
    upright_anchors: tuple
    preserve_attachment_handedness: bool = True

    @property
    def inboard_point(self) -> PointID:
        return P.TRACKROD_INBOARD if self.steered else P.TOE_LINK_INBOARD

    @property
    def outboard_point(self) -> PointID:
        return P.TRACKROD_OUTBOARD if self.steered else P.TOE_LINK_OUTBOARD

    @property
    def required_points(self) -> frozenset:
        return frozenset({self.inboard_point, self.outboard_point})

    @property
    def free_points(self) -> tuple:
        return (self.outboard_point, self.inboard_point) if self.steered else (self.outboard_point,)

    @property
    def output_points(self) -> tuple:
        return (self.inboard_point, self.outboard_point)

    def constraints(self, positions) -> list[Constraint]:
        inboard, outboard = self.inboard_point, self.outboard_point
        if self.preserve_attachment_handedness:
            attach = anchored_rigid_point_constraints(positions, outboard, self.upright_anchors)
        else:
            attach = [_distance(positions, outboard, anchor) for anchor in self.upright_anchors]
        rows: list[Constraint] = [_distance(positions, inboard, outboard), *attach]
        if self.steered:  # rack translation: the pickup stays on the world-Y line through its design position
            rows.append(PointOnLineConstraint(inboard, positions[inboard].data.copy(), WORLD_Y))
        return rows


ROCKER_AXIS = (P.ROCKER_AXIS_A, P.ROCKER_AXIS_B)
ROCKER_BODY = (P.ROCKER_AXIS_A, P.ROCKER_AXIS_B, P.PUSHROD_INBOARD)


@dataclass(frozen=True)
class Actuation:
    """``direct`` or ``pushrod_rocker`` (``corner/mechanisms.py:80-302``)."""

    kind: str
    body: tuple  # rigid body carrying the moving pickup
    external_pickups: tuple = ()  # rocker pickups owned by axle hardware (droplink ...)

    @property
    def rocker(self) -> bool:
        return self.kind == "pushrod_rocker"

    @property
    def required_points(self) -> frozenset:
        if not self.rocker:
            return frozenset()
        return frozenset({P.PUSHROD_OUTBOARD, P.PUSHROD_INBOARD, *ROCKER_AXIS, *self.external_pickups})

    @property
    def free_points(self) -> tuple:
        return (P.PUSHROD_OUTBOARD, P.PUSHROD_INBOARD, *self.external_pickups) if self.rocker else ()

    output_points = free_points

    def validate(self, hardpoints) -> None:
        label = "Pushrod actuation" if self.rocker else "Direct spring actuation"
        validate_rigid_anchor_points(hardpoints, self.body, label)
        if not self.rocker:
            return
        a, b = hardpoints[P.ROCKER_AXIS_A].data, hardpoints[P.ROCKER_AXIS_B].data
        if float(np.linalg.norm(b - a)) <= EPS_GEOMETRIC:
            raise ValueError("Rocker axis points must be distinct")
        axis = (b - a) / float(np.linalg.norm(b - a))
        for point in (P.PUSHROD_INBOARD, *self.external_pickups):
            if float(np.linalg.norm(np.cross(hardpoints[point].data - a, axis))) <= EPS_GEOMETRIC:
                raise ValueError(f"{point.name} must not lie on the rocker axis")

    def constraints(self, positions) -> list[Constraint]:
        if not self.rocker:
            return []
        rows = anchored_rigid_point_constraints(positions, P.PUSHROD_OUTBOARD, self.body)
        rows += [
            _distance(positions, P.PUSHROD_OUTBOARD, P.PUSHROD_INBOARD),
            _distance(positions, P.PUSHROD_INBOARD, P.ROCKER_AXIS_A),
            _distance(positions, P.PUSHROD_INBOARD, P.ROCKER_AXIS_B),
        ]
        for point in self.external_pickups:
            rows += chiral_rigid_point_constraints(positions, point, ROCKER_BODY)
        return rows

    def spring_constraints(self, positions) -> list[Constraint]:
        if self.rocker:
            return chiral_rigid_point_constraints(positions, P.STRUT_BOTTOM, ROCKER_BODY)
        return anchored_rigid_point_constraints(positions, P.STRUT_BOTTOM, self.body)


@dataclass(frozen=True)
class CornerSpring:
    """``none``, ``coilover`` or ``torsion_bar`` (``corner/mechanisms.py:436-642``)."""

    kind: str

    @property
    def coilover(self) -> bool:
        return self.kind == "coilover"

    @property
    def required_points(self) -> frozenset:
        return frozenset({P.STRUT_TOP, P.STRUT_BOTTOM}) if self.coilover else frozenset()

    @property
    def free_points(self) -> tuple:
        return (P.STRUT_BOTTOM,) if self.coilover else ()

    @property
    def output_points(self) -> tuple:
        return (P.STRUT_TOP, P.STRUT_BOTTOM) if self.coilover else ()

    def validate(self, actuation: Actuation) -> None:
        if self.kind == "torsion_bar" and not actuation.rocker:
            raise ValueError("Corner torsion bar is not supported by direct actuation yet")

    def constraints(self, positions, actuation: Actuation) -> list[Constraint]:
        return actuation.spring_constraints(positions) if self.coilover else []


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class WheelConfig:
    offset: float
    section_width: float
    aspect_ratio: float
    rim_diameter: float  # inches

    @property
    def nominal_radius(self) -> float:
        """``schema/config.py:28-41``."""
        return (self.rim_diameter * MM_PER_INCH + 2 * (self.aspect_ratio * self.section_width)) / 2


@dataclass(frozen=True)
class VehicleSetup:
    """Vehicle-wide numbers the anti-geometry metrics read (``schema/config.py:72-90,140``)."""

    wheelbase: float
    cg_position: tuple
    front_brake_bias: float | None = None
    driven_axle: str | None = None    # "front" / "rear"
    axle_position: str | None = None  # "front" / "rear"

    def __post_init__(self) -> None:
        if self.front_brake_bias is not None and not 0.0 <= self.front_brake_bias <= 1.0:
            raise ValueError(f"front_brake_bias must be in [0, 1], got {self.front_brake_bias}")
        for value in (self.driven_axle, self.axle_position):
            if value not in (None, "front", "rear"):
                raise ValueError(f"axle position must be 'front' or 'rear', got {value!r}")


WHEEL_OUTPUT_POINTS = (
    P.AXLE_INBOARD, P.AXLE_OUTBOARD, P.AXLE_MIDPOINT, P.WHEEL_CENTER,
    P.WHEEL_INBOARD, P.WHEEL_OUTBOARD, P.CONTACT_PATCH_CENTER,
)


class Suspension:
    """The protocol ``solve_sweep`` needs (reference ``suspensions/base.py:88-126,227-244``)."""

    def initial_state(self) -> SuspensionState:
        raise NotImplementedError

    def constraints(self) -> list[Constraint]:
        raise NotImplementedError

    def derived_spec(self) -> DerivedPointsSpec:
        raise NotImplementedError

    def free_points(self) -> Sequence[Any]:
        raise NotImplementedError

    def output_points(self) -> tuple:
        raise NotImplementedError

    def actuator_dofs(self) -> tuple:
        raise NotImplementedError

    def resolve_target_key(self, point: PointID, side):
        raise NotImplementedError


@dataclass
class CornerSuspension(Suspension):
    name: str
    side: Side
    hardpoints: dict
    wheel: WheelConfig
    steered: bool
    vehicle: VehicleSetup | None = None
    _state: SuspensionState | None = field(default=None, init=False, repr=False)

    REQUIRED = frozenset()

    def required_points(self) -> frozenset:
        return self.REQUIRED

    def validate_hardpoints(self) -> None:
        missing = self.required_points() - set(self.hardpoints)
        if missing:
            raise ValueError("Missing required hardpoints: " + ", ".join(sorted(p.name for p in missing)))
        extra = set(self.hardpoints) - self.required_points()
        if extra:
            raise ValueError("Unexpected hardpoints for this topology: " + ", ".join(sorted(p.name for p in extra)))
        outboard_y = float(self.hardpoints[P.AXLE_OUTBOARD].data[1])
        if self.side is Side.LEFT and outboard_y <= 0.0:
            raise ValueError(f"Side 'left' requires AXLE_OUTBOARD Y > 0 (got {outboard_y}); check the hardpoint handedness.")
        if self.side is Side.RIGHT and outboard_y >= 0.0:
            raise ValueError(f"Side 'right' requires AXLE_OUTBOARD Y < 0 (got {outboard_y}); check the hardpoint handedness.")

    def rack_attachment_point(self):
        return self.heading_link.inboard_point if self.steered else None

    def wheel_axis_points(self) -> tuple:
        """Wheel spin axis, inboard -> outboard (``corner/base.py`` role hook)."""
        return (P.AXLE_INBOARD, P.AXLE_OUTBOARD)

    def steering_axis_points(self) -> tuple:
        """(lower, upper) steering pivots; architecture specific."""
        raise NotImplementedError

    def damper_points(self):
        """(top, bottom) of an installed linear spring/damper, or None (``suspensions/base.py:188-190``)."""
        return None

    def instant_axis_points(self):
        """(kind, points) the upright's instant axis is constructed from, or None (``compute_instant_axis``)."""
        return None

    @property
    def lateral_sign(self) -> float:
        """``Side.lateral_sign``: +1 left, -1 right."""
        return 1.0 if self.side is Side.LEFT else -1.0

    def actuator_dofs(self) -> tuple:
        """``corner/base.py:80-91``."""
        point = self.rack_attachment_point()
        if point is None:
            return ()
        return (ActuatorDOF("steering rack", (point,), WORLD_Y),)

    def resolve_target_key(self, point: PointID, side):
        if side is not None and side is not self.side:
            raise ValueError(f"Sweep target side '{side.name.lower()}' does not match this corner")
        return point

    def initial_state(self) -> SuspensionState:
        if self._state is None:
            positions = {k: v.copy() for k, v in self.hardpoints.items()}
            self.apply_setup(positions)
            DerivedPointsManager(self.derived_spec()).update_in_place(positions)
            self._state = SuspensionState(positions=positions, free_points=set(self.free_points()))
        return self._state

    def apply_setup(self, positions: dict) -> None:
        """Setup changes applied to the authored hardpoints before the design state is built (shims)."""

    def wheel_spec(self) -> DerivedPointsSpec:
        return build_wheel_derived_spec(self.wheel.offset, self.wheel.section_width, self.wheel.nominal_radius)


@dataclass
class DoubleWishboneSuspension(CornerSuspension):
    """``corner/double_wishbone.py``."""

    actuation: Actuation = None  # type: ignore[assignment]
    spring: CornerSpring = CornerSpring("none")
    camber_shim: Any = None  # shims.CamberShimConfig or None

    REQUIRED = frozenset({
        P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD,
        P.UPPER_WISHBONE_INBOARD_FRONT, P.UPPER_WISHBONE_INBOARD_REAR, P.UPPER_WISHBONE_OUTBOARD,
        P.AXLE_INBOARD, P.AXLE_OUTBOARD,
    })
    LOWER_WISHBONE_BODY = (P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD)
    UPRIGHT_BODY = (P.UPPER_WISHBONE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD, P.AXLE_INBOARD, P.AXLE_OUTBOARD)
    MOUNT_BODIES = {"lower_wishbone": LOWER_WISHBONE_BODY, "upright": UPRIGHT_BODY}
    LOCATING_OUTPUT_POINTS = (
        P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD,
        P.UPPER_WISHBONE_INBOARD_FRONT, P.UPPER_WISHBONE_INBOARD_REAR, P.UPPER_WISHBONE_OUTBOARD,
    )
    FREE_POINTS = (P.UPPER_WISHBONE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD, P.AXLE_INBOARD, P.AXLE_OUTBOARD)
    LENGTH_PAIRS = (
        (P.UPPER_WISHBONE_INBOARD_FRONT, P.UPPER_WISHBONE_OUTBOARD),
        (P.UPPER_WISHBONE_INBOARD_REAR, P.UPPER_WISHBONE_OUTBOARD),
        (P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_OUTBOARD),
        (P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD),
        (P.UPPER_WISHBONE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD),
        (P.AXLE_INBOARD, P.AXLE_OUTBOARD),
        (P.AXLE_INBOARD, P.UPPER_WISHBONE_OUTBOARD),
        (P.AXLE_INBOARD, P.LOWER_WISHBONE_OUTBOARD),
        (P.AXLE_OUTBOARD, P.UPPER_WISHBONE_OUTBOARD),
        (P.AXLE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD),
    )

    def __post_init__(self) -> None:
        # the four upright anchors already overdetermine the pickup; the upright angle row
        # keeps the authored branch (double_wishbone.py:163-181)
        self.heading_link = HeadingLink(self.steered, self.UPRIGHT_BODY, preserve_attachment_handedness=False)
        if self.actuation is None:
            self.actuation = Actuation("direct", self.LOWER_WISHBONE_BODY)
        self.validate_hardpoints()
        self.actuation.validate(self.hardpoints)
        self.spring.validate(self.actuation)
        validate_rigid_anchor_points(self.hardpoints, self.UPRIGHT_BODY, "Track rod" if self.steered else "Toe link")

    def required_points(self) -> frozenset:
        return self.REQUIRED | self.heading_link.required_points | self.actuation.required_points | self.spring.required_points

    def steering_axis_points(self) -> tuple:
        """``double_wishbone.py:223-225``: the two outboard ball joints."""
        return (P.LOWER_WISHBONE_OUTBOARD, P.UPPER_WISHBONE_OUTBOARD)

    def upright_attachment_points(self) -> tuple:
        """Points the upright carries through a camber-shim change (``double_wishbone.py:572-581``)."""
        base = (P.AXLE_INBOARD, P.AXLE_OUTBOARD, self.heading_link.outboard_point)
        if self.actuation.body == self.UPRIGHT_BODY:
            pickup = P.PUSHROD_OUTBOARD if self.actuation.rocker else P.STRUT_BOTTOM
            if pickup in self.hardpoints:
                return (*base, pickup)
        return base

    def shim_rocker_points(self):
        """Rocker group an upright-mounted pushrod turns during the shim solve, else None
        (``double_wishbone.py:517-533,564-569``, ``mechanisms.py:198-200,247-265``)."""
        if not (self.actuation.rocker and self.actuation.body == self.UPRIGHT_BODY):
            return None
        spring_points = (P.STRUT_BOTTOM,) if self.spring.coilover else ()
        return tuple(dict.fromkeys((P.PUSHROD_INBOARD, *self.actuation.external_pickups, *spring_points)))

    def apply_setup(self, positions: dict) -> None:
        """``double_wishbone.py:232-244,501-570``: the camber-shim setup solve (on the device)."""
        if self.camber_shim is not None and not self.camber_shim.unchanged:
            from .shims import apply_camber_shim

            apply_camber_shim(self, positions)

    def damper_points(self):
        """``double_wishbone.py:219-221``: the coil-over's mounts when one is installed."""
        return (P.STRUT_TOP, P.STRUT_BOTTOM) if self.spring.coilover else None

    def instant_axis_points(self):
        """``double_wishbone.py:376-403``: where the upper and lower wishbone planes meet."""
        return ("two_planes", (P.UPPER_WISHBONE_INBOARD_FRONT, P.UPPER_WISHBONE_INBOARD_REAR, P.UPPER_WISHBONE_OUTBOARD,
                               P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD))

    def free_points(self) -> tuple:
        return (*self.FREE_POINTS, *self.heading_link.free_points, *self.actuation.free_points, *self.spring.free_points)

    def output_points(self) -> tuple:
        return tuple(dict.fromkeys((
            *self.LOCATING_OUTPUT_POINTS, *self.heading_link.output_points, *WHEEL_OUTPUT_POINTS,
            *self.actuation.output_points, *self.spring.output_points,
        )))

    def derived_spec(self) -> DerivedPointsSpec:
        return self.wheel_spec()

    def constraints(self) -> list[Constraint]:
        pos = self.initial_state().positions
        rows: list[Constraint] = [_distance(pos, a, b) for a, b in self.LENGTH_PAIRS]
        v1 = pos[P.LOWER_WISHBONE_OUTBOARD].data - pos[P.UPPER_WISHBONE_OUTBOARD].data
        v2 = pos[P.AXLE_OUTBOARD].data - pos[P.AXLE_INBOARD].data
        rows.append(AngleConstraint(P.UPPER_WISHBONE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD,
                                    P.AXLE_INBOARD, P.AXLE_OUTBOARD, _angle(v1, v2)))
        rows += self.heading_link.constraints(pos)
        rows += self.actuation.constraints(pos)
        rows += self.spring.constraints(pos, self.actuation)
        return rows


STRUT_AXIS_ALIGNMENT_TOLERANCE_MM = 1.0


@dataclass
class MacPhersonSuspension(CornerSuspension):
    """``corner/macpherson.py``: strut clamp derived on the ball-joint-to-top-mount line."""

    REQUIRED = frozenset({
        P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD,
        P.STRUT_TOP, P.STRUT_BOTTOM, P.AXLE_INBOARD, P.AXLE_OUTBOARD,
    })
    UPRIGHT_BODY = (P.LOWER_WISHBONE_OUTBOARD, P.AXLE_INBOARD, P.AXLE_OUTBOARD)
    LOCATING_OUTPUT_POINTS = (
        P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD,
        P.STRUT_TOP, P.STRUT_BOTTOM,
    )
    FREE_POINTS = (P.LOWER_WISHBONE_OUTBOARD, P.AXLE_INBOARD, P.AXLE_OUTBOARD)
    LENGTH_PAIRS = (
        (P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_OUTBOARD),
        (P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD),
        (P.AXLE_INBOARD, P.AXLE_OUTBOARD),
        (P.AXLE_INBOARD, P.LOWER_WISHBONE_OUTBOARD),
        (P.AXLE_OUTBOARD, P.LOWER_WISHBONE_OUTBOARD),
    )

    def __post_init__(self) -> None:
        self.heading_link = HeadingLink(self.steered, self.UPRIGHT_BODY)
        self.validate_hardpoints()
        validate_rigid_anchor_points(self.hardpoints, self.UPRIGHT_BODY, "Track rod" if self.steered else "Toe link")
        ball, top = self.hardpoints[P.LOWER_WISHBONE_OUTBOARD].data, self.hardpoints[P.STRUT_TOP].data
        axis_length = float(np.linalg.norm(top - ball))
        if axis_length <= EPS_GEOMETRIC:
            raise ValueError("STRUT_TOP must not coincide with LOWER_WISHBONE_OUTBOARD; the steering axis would be undefined.")
        axis = (top - ball) / axis_length
        clamp = self.hardpoints[P.STRUT_BOTTOM].data - ball
        offset = float(np.linalg.norm(np.cross(clamp, axis)))
        if offset > STRUT_AXIS_ALIGNMENT_TOLERANCE_MM:
            raise ValueError(
                f"STRUT_BOTTOM sits {offset:.3f} mm off the line from LOWER_WISHBONE_OUTBOARD to STRUT_TOP. "
                "This model treats the strut axis as coincident with the steering axis; "
                "an intentionally offset strut is not supported."
            )
        axial = self._strut_clamp_offset()
        if axial <= EPS_GEOMETRIC or axial >= axis_length - EPS_GEOMETRIC:
            raise ValueError("STRUT_BOTTOM must lie between LOWER_WISHBONE_OUTBOARD and STRUT_TOP along the strut axis")

    def _strut_clamp_offset(self) -> float:
        """Authored ball-joint-to-clamp distance along the strut axis (``macpherson.py:199-204``)."""
        ball = self.hardpoints[P.LOWER_WISHBONE_OUTBOARD].data
        v = self.hardpoints[P.STRUT_TOP].data - ball
        axis = v / float(np.linalg.norm(v))
        return float((self.hardpoints[P.STRUT_BOTTOM].data - ball).dot(axis))

    def required_points(self) -> frozenset:
        return self.REQUIRED | self.heading_link.required_points

    def steering_axis_points(self) -> tuple:
        """``macpherson.py:210-212``: lower ball joint to the strut top."""
        return (P.LOWER_WISHBONE_OUTBOARD, P.STRUT_TOP)

    def damper_points(self):
        """``macpherson.py:220-222``: the strut is the spring/damper."""
        return (P.STRUT_TOP, P.STRUT_BOTTOM)

    def instant_axis_points(self):
        """``macpherson.py:325-355``: lower-arm plane and the plane through the strut top normal to the strut."""
        return ("plane_and_strut", (P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR,
                                    P.LOWER_WISHBONE_OUTBOARD, P.STRUT_TOP))

    def free_points(self) -> tuple:
        return (*self.FREE_POINTS, *self.heading_link.free_points)

    def output_points(self) -> tuple:
        return tuple(dict.fromkeys((*self.LOCATING_OUTPUT_POINTS, *self.heading_link.output_points, *WHEEL_OUTPUT_POINTS)))

    def derived_spec(self) -> DerivedPointsSpec:
        wheel = self.wheel_spec()
        functions = {
            P.STRUT_BOTTOM: partial(get_point_along_line, start_point=P.LOWER_WISHBONE_OUTBOARD,
                                    end_point=P.STRUT_TOP, distance_from_start=self._strut_clamp_offset()),
            **wheel.functions,
        }
        dependencies = {P.STRUT_BOTTOM: {P.LOWER_WISHBONE_OUTBOARD, P.STRUT_TOP}, **wheel.dependencies}
        return DerivedPointsSpec(functions, dependencies)

    def constraints(self) -> list[Constraint]:
        pos = self.initial_state().positions
        rows: list[Constraint] = [_distance(pos, a, b) for a, b in self.LENGTH_PAIRS]
        rows += chiral_rigid_point_constraints(pos, P.STRUT_BOTTOM, self.UPRIGHT_BODY)
        rows += self.heading_link.constraints(pos)
        return rows


# --------------------------------------------------------------------------------------
# axle
# --------------------------------------------------------------------------------------


class _SideView:
    """One side of a PointRef-keyed position map seen with PointID keys."""

    def __init__(self, positions, side: Side):
        self._positions, self._side = positions, side

    def __getitem__(self, point):
        return self._positions[PointRef(self._side, point)]


def _wrap_side(function, side: Side):
    def wrapped(positions):
        return function(_SideView(positions, side))

    wrapped.__okx_inner__ = function  # lets program.flatten_problem see through the wrapper
    wrapped.__okx_side__ = side
    return wrapped


@dataclass
class AxleSuspension(Suspension):
    """Two corners + rack coupling + the shared hardware: a U-bar or a rigid T-bar anti-roll bar and a
    rocker-to-rocker heave link (``axle/suspension.py``, ``axle/mechanisms.py:228-342,600-720,880-960``)."""

    name: str
    corners: dict
    arb_center_points: dict = field(default_factory=dict)      # PointID -> Point3 (U-bar axis / T-bar pivot)
    arb_droplink_points: dict = field(default_factory=dict)    # Side -> Point3
    arb_kind: str = ""                                         # "", "u_bar", "t_bar" ("" = from the points given)
    heave_link: bool = False                                   # variable-length link between the rockers
    _state: SuspensionState | None = field(default=None, init=False, repr=False)

    def __post_init__(self) -> None:
        if set(self.corners) != {Side.LEFT, Side.RIGHT}:
            raise ValueError("Axle requires exactly LEFT and RIGHT corner models.")
        self.rack_attachment_points()
        if not self.arb_kind and self.has_arb:
            self.arb_kind = "t_bar" if P.ARB_T_BAR_PIVOT in self.arb_center_points else "u_bar"
        if self.arb_kind == "u_bar":
            if set(self.arb_center_points) != {P.ARB_U_BAR_AXIS_A, P.ARB_U_BAR_AXIS_B}:
                raise ValueError("U-bar requires center ARB_U_BAR_AXIS_A and ARB_U_BAR_AXIS_B")
            if set(self.arb_droplink_points) != {Side.LEFT, Side.RIGHT}:
                raise ValueError("U-bar requires DROPLINK_U_BAR on both sides")
            for side, corner in self.corners.items():
                if P.DROPLINK_ROCKER not in corner.free_points():
                    raise ValueError(f"{side.name} U-bar corner does not expose DROPLINK_ROCKER as a moving pickup")
        elif self.arb_kind == "t_bar":
            self._validate_t_bar()
        if self.heave_link:  # mechanisms.py:884-899
            for side, corner in self.corners.items():
                if P.HEAVE_LINK_ROCKER not in corner.free_points():
                    raise ValueError(f"{side.name} corner does not expose HEAVE_LINK_ROCKER as a moving pickup")
            left = self.corners[Side.LEFT].initial_state().positions[P.HEAVE_LINK_ROCKER].data
            right = self.corners[Side.RIGHT].initial_state().positions[P.HEAVE_LINK_ROCKER].data
            if float(np.linalg.norm(left - right)) <= EPS_GEOMETRIC:
                raise ValueError("Rocker-to-rocker heave-link pickups must be separated in the design state")

    def _validate_t_bar(self) -> None:
        """``ArbTBar.validate`` (``axle/mechanisms.py:610-646``): pivot and crossbar midpoint on the centre line,
        a proper triangle."""
        for side, corner in self.corners.items():
            if P.DROPLINK_ROCKER not in corner.free_points():
                raise ValueError(f"{side.name} T-bar corner does not expose DROPLINK_ROCKER as a moving pickup")
        if set(self.arb_center_points) != {P.ARB_T_BAR_PIVOT}:
            raise ValueError("T-bar requires center ARB_T_BAR_PIVOT")
        if set(self.arb_droplink_points) != {Side.LEFT, Side.RIGHT}:
            raise ValueError("T-bar requires DROPLINK_T_BAR on both sides")
        pivot = self.arb_center_points[P.ARB_T_BAR_PIVOT].data
        if abs(float(pivot[1])) > EPS_GEOMETRIC:
            raise ValueError("ARB_T_BAR_PIVOT must lie on the vehicle centerline Y = 0")
        left, right = self.arb_droplink_points[Side.LEFT].data, self.arb_droplink_points[Side.RIGHT].data
        middle = left + (right - left) / 2.0
        if abs(float(middle[1])) > EPS_GEOMETRIC:
            raise ValueError("The T-bar crossbar midpoint must lie on the vehicle centerline Y = 0")
        crossbar, stem = right - left, middle - pivot
        if float(np.linalg.norm(crossbar)) <= EPS_GEOMETRIC:
            raise ValueError("T-bar crossbar points must be distinct")
        if float(np.linalg.norm(stem)) <= EPS_GEOMETRIC:
            raise ValueError("T-bar pivot and crossbar midpoint must be distinct")
        if float(np.linalg.norm(np.cross(crossbar, stem))) <= EPS_GEOMETRIC:
            raise ValueError("T-bar points must define a non-degenerate triangle")

    @property
    def has_arb(self) -> bool:
        return bool(self.arb_center_points) or bool(self.arb_droplink_points)

    @property
    def arb_arm_point(self):
        """The moving ARB pickup a droplink ends on, per side: ``DROPLINK_U_BAR`` / ``DROPLINK_T_BAR`` / None."""
        return {"u_bar": P.DROPLINK_U_BAR, "t_bar": P.DROPLINK_T_BAR}.get(self.arb_kind)

    def rack_attachment_points(self):
        left = self.corners[Side.LEFT].rack_attachment_point()
        right = self.corners[Side.RIGHT].rack_attachment_point()
        if (left is None) != (right is None):
            raise ValueError("Axle corners disagree on rack attachment: one corner is steered and the other is not.")
        return None if left is None else (left, right)

    def actuator_dofs(self) -> tuple:
        rack = self.rack_attachment_points()
        if rack is None:
            return ()
        return (ActuatorDOF("steering rack", (PointRef(Side.LEFT, rack[0]), PointRef(Side.RIGHT, rack[1])), WORLD_Y),)

    def resolve_target_key(self, point: PointID, side):
        if side not in (Side.LEFT, Side.RIGHT):
            raise ValueError(f"Axle sweep target for '{point.name}' requires side left or right.")
        return PointRef(side, point)

    def _arb_free(self) -> tuple:
        arm = self.arb_arm_point
        if arm is None:
            return ()
        return (PointRef(Side.LEFT, arm), PointRef(Side.RIGHT, arm))

    def initial_state(self) -> SuspensionState:
        if self._state is None:
            positions: dict = {}
            free: set = set()
            for side, corner in self.corners.items():
                state = corner.initial_state()
                positions.update({PointRef(side, k): v.copy() for k, v in state.positions.items()})
                free.update(PointRef(side, k) for k in state.free_points)
            for point, position in self.arb_center_points.items():
                positions[PointRef(Side.CENTER, point)] = position.copy()
            for side, position in self.arb_droplink_points.items():
                key = PointRef(side, self.arb_arm_point)
                positions[key] = position.copy()
                free.add(key)
            self._state = SuspensionState(positions, free)
        return self._state

    def free_points(self) -> tuple:
        corner = tuple(PointRef(s, p) for s, c in self.corners.items() for p in c.free_points())
        return (*corner, *self._arb_free())

    def output_points(self) -> tuple:
        corner = tuple(PointRef(s, p) for s in (Side.LEFT, Side.RIGHT) for p in self.corners[s].output_points())
        return tuple(dict.fromkeys((*corner, *self._arb_free())))

    def constraints(self) -> list[Constraint]:
        rows = [c.remap(lambda p, side=side: PointRef(side, p))
                for side, corner in self.corners.items() for c in corner.constraints()]
        rack = self.rack_attachment_points()
        if rack is not None:  # rigid rack: fixed distance between the two pickups (suspension.py:196-209)
            left = self.corners[Side.LEFT].initial_state().positions[rack[0]].data
            right = self.corners[Side.RIGHT].initial_state().positions[rack[1]].data
            rows.append(DistanceConstraint(PointRef(Side.LEFT, rack[0]), PointRef(Side.RIGHT, rack[1]),
                                           float(np.linalg.norm(right - left))))
        rocker = {side: self.corners[side].initial_state().positions.get(P.DROPLINK_ROCKER) for side in self.corners}
        if self.arb_kind == "u_bar":  # mechanisms.py:307-342
            axis_a = self.arb_center_points[P.ARB_U_BAR_AXIS_A].data
            axis_b = self.arb_center_points[P.ARB_U_BAR_AXIS_B].data
            key_a, key_b = PointRef(Side.CENTER, P.ARB_U_BAR_AXIS_A), PointRef(Side.CENTER, P.ARB_U_BAR_AXIS_B)
            for side in (Side.LEFT, Side.RIGHT):
                drop = self.arb_droplink_points[side].data
                arm = PointRef(side, P.DROPLINK_U_BAR)
                rows += [
                    DistanceConstraint(arm, key_a, float(np.linalg.norm(axis_a - drop))),
                    DistanceConstraint(arm, key_b, float(np.linalg.norm(axis_b - drop))),
                    DistanceConstraint(PointRef(side, P.DROPLINK_ROCKER), arm, float(np.linalg.norm(drop - rocker[side].data))),
                ]
        elif self.arb_kind == "t_bar":
            # rigid triangle (two crossbar ends + chassis pivot), crossbar midpoint on the vehicle XZ plane, one
            # droplink per side (mechanisms.py:670-716); the heave link adds no row - its length is free
            pivot_key = PointRef(Side.CENTER, P.ARB_T_BAR_PIVOT)
            pivot = self.arb_center_points[P.ARB_T_BAR_PIVOT].data
            ends = {side: self.arb_droplink_points[side].data for side in (Side.LEFT, Side.RIGHT)}
            arm = {side: PointRef(side, P.DROPLINK_T_BAR) for side in (Side.LEFT, Side.RIGHT)}
            rows += [
                DistanceConstraint(arm[Side.LEFT], arm[Side.RIGHT], float(np.linalg.norm(ends[Side.LEFT] - ends[Side.RIGHT]))),
                DistanceConstraint(arm[Side.LEFT], pivot_key, float(np.linalg.norm(ends[Side.LEFT] - pivot))),
                DistanceConstraint(arm[Side.RIGHT], pivot_key, float(np.linalg.norm(ends[Side.RIGHT] - pivot))),
                MidpointOnPlaneConstraint(arm[Side.LEFT], arm[Side.RIGHT], np.zeros(3), WORLD_Y),
            ]
            for side in (Side.LEFT, Side.RIGHT):
                rows.append(DistanceConstraint(PointRef(side, P.DROPLINK_ROCKER), arm[side],
                                               float(np.linalg.norm(rocker[side].data - ends[side]))))
        return rows

    def derived_spec(self) -> DerivedPointsSpec:
        functions, dependencies = {}, {}
        for side, corner in self.corners.items():
            spec = corner.derived_spec()
            for point, function in spec.functions.items():
                functions[PointRef(side, point)] = _wrap_side(function, side)
                dependencies[PointRef(side, point)] = {PointRef(side, d) for d in spec.dependencies[point]}
        return DerivedPointsSpec(functions, dependencies)

    def corner_state(self, state: SuspensionState, side: Side) -> SuspensionState:
        positions = {k.point: v for k, v in state.positions.items() if isinstance(k, PointRef) and k.side is side}
        free = {k.point for k in state.free_points if isinstance(k, PointRef) and k.side is side}
        return SuspensionState(positions, free)
