"""
YAML / mapping front end: geometry -> suspension model, sweep mapping -> ``SweepConfig``
(reference ``core/input.py``, ``core/schema/sweep.py:137-196``, ``suspensions/build.py``,
``cli/io/loaders.py``, ``cli/io/sweep_loader.py``).  Covers the fields the BASELINE
configurations use; unknown keys are rejected like the reference's ``extra="forbid"`` models.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Mapping

import numpy as np

from .enums import Axis, PointID, Side, TargetPositionMode
from .shims import CamberShimConfig
from .state import Point3
from .targeting import PointTarget, PointTargetAxis, PointTargetVector, SweepConfig, validate_sweep_controls
from .topology import (
    Actuation,
    AxleSuspension,
    CornerSpring,
    DoubleWishboneSuspension,
    MacPhersonSuspension,
    Suspension,
    VehicleSetup,
    WheelConfig,
)

_AXES = {"x": Axis.X, "y": Axis.Y, "z": Axis.Z}


def _require_keys(data: Mapping[str, Any], allowed: set, what: str) -> None:
    extra = set(data) - allowed
    if extra:
        raise ValueError(f"Invalid {what} specification: unexpected keys {sorted(extra)}")


def _point_id(name: str) -> PointID:
    try:
        return PointID[str(name).upper()]
    except KeyError as error:
        raise ValueError(f"Unknown point name: '{name}'") from error


def _side(value) -> Side:
    if isinstance(value, Side):
        return value
    try:
        return Side[str(value).upper()]
    except KeyError as error:
        raise ValueError(f"Unknown side: '{value}'") from error


def _hardpoints(data: Mapping[str, Any]) -> dict:
    points = {}
    for name, xyz in data.items():
        _require_keys(xyz, {"x", "y", "z"}, f"hardpoint '{name}'")
        points[_point_id(name)] = Point3([float(xyz["x"]), float(xyz["y"]), float(xyz["z"])])
    return points


def _wheel(config: Mapping[str, Any]) -> WheelConfig:
    wheel = config["wheel"]
    tire = wheel["tire"]
    return WheelConfig(
        offset=float(wheel.get("offset", 0.0)),
        section_width=float(tire["section_width"]),
        aspect_ratio=float(tire["aspect_ratio"]),
        rim_diameter=float(tire["rim_diameter"]),
    )


def _vehicle(vehicle: Mapping[str, Any], axle_position) -> VehicleSetup | None:
    """``VehicleConfig`` + the axle's position (``schema/config.py:72-90,120,140``); None when not authored."""
    if vehicle.get("wheelbase") is None or vehicle.get("cg_position") is None:
        return None
    cg = vehicle["cg_position"]
    bias = vehicle.get("front_brake_bias")
    lower = lambda v: None if v is None else str(v).lower()  # noqa: E731
    return VehicleSetup(
        wheelbase=float(vehicle["wheelbase"]),
        cg_position=(float(cg["x"]), float(cg["y"]), float(cg["z"])),
        front_brake_bias=None if bias is None else float(bias),
        driven_axle=lower(vehicle.get("driven_axle")),
        axle_position=lower(axle_position),
    )


def _steered(config: Mapping[str, Any]) -> bool:
    kind = str(config.get("steering", {}).get("type", "none")).lower()
    if kind not in ("rack", "none"):
        raise ValueError(f"Unsupported steering type: '{kind}'")
    return kind == "rack"


def _shim(config: Mapping[str, Any], kind: str) -> CamberShimConfig | None:
    """``CamberShimConfig`` (``schema/config.py:52-70``); only the double wishbone supports it (``build.py:378-391``)."""
    shim = config.get("camber_shim") if config else None
    if shim is None:
        return None
    if kind == "macpherson":
        raise ValueError(f"Suspension type '{kind}' does not support outboard camber shims")
    if kind != "double_wishbone":
        return None  # unknown type: _build_corner reports it
    _require_keys(shim, {"shim_face_point_a", "shim_face_point_b", "shim_face_normal", "design_thickness",
                         "setup_thickness"}, "camber_shim")
    xyz = lambda v: (float(v["x"]), float(v["y"]), float(v["z"]))  # noqa: E731
    return CamberShimConfig(xyz(shim["shim_face_point_a"]), xyz(shim["shim_face_point_b"]), xyz(shim["shim_face_normal"]),
                            float(shim["design_thickness"]), float(shim["setup_thickness"]))


def _mechanisms(actuation: Mapping[str, Any] | None, spring: Mapping[str, Any] | None,
                external_pickups: tuple = ()):
    actuation = actuation or {"type": "direct", "mount": "lower_wishbone"}
    spring = spring or {"type": "none"}
    kind, mount = str(actuation["type"]).lower(), str(actuation["mount"]).lower()
    if kind not in ("direct", "pushrod_rocker"):
        raise ValueError(f"Unsupported actuation type: {kind}")
    if mount not in DoubleWishboneSuspension.MOUNT_BODIES:
        raise ValueError(f"Architecture does not provide the '{mount}' mounting body")
    if kind == "direct" and external_pickups:
        raise ValueError("Direct actuation does not accept rocker pickups")
    spring_kind = str(spring["type"]).lower()
    if spring_kind not in ("none", "coilover", "torsion_bar"):
        raise ValueError(f"Unsupported corner spring type: {spring_kind}")
    if kind == "direct" and spring_kind == "torsion_bar":
        raise ValueError("Direct torsion-bar actuation is not implemented yet")
    return (Actuation(kind, DoubleWishboneSuspension.MOUNT_BODIES[mount], tuple(external_pickups)),
            CornerSpring(spring_kind))


def _mirror(points: dict) -> dict:
    """Reflect through the vehicle XZ plane (``build.py:344-354``)."""
    return {k: Point3([float(v.data[0]), -float(v.data[1]), float(v.data[2])]) for k, v in points.items()}


def _build_corner(kind: str, name: str, side: Side, hardpoints: dict, config: Mapping[str, Any],
                  actuation=None, spring=None, external_pickups: tuple = (), vehicle: VehicleSetup | None = None,
                  camber_shim: CamberShimConfig | None = None):
    if kind == "double_wishbone":
        act, spr = _mechanisms(actuation, spring, external_pickups)
        return DoubleWishboneSuspension(name=name, side=side, hardpoints=hardpoints, wheel=_wheel(config),
                                        steered=_steered(config), vehicle=vehicle, actuation=act, spring=spr,
                                        camber_shim=camber_shim)
    if kind == "macpherson":
        return MacPhersonSuspension(name=name, side=side, hardpoints=hardpoints, wheel=_wheel(config),
                                    steered=_steered(config), vehicle=vehicle)
    raise ValueError(f"Unsupported geometry type: '{kind}'")


def build_suspension(data: Mapping[str, Any]) -> Suspension:
    """Decode, validate and build one suspension mapping (``core/input.py:57-60``)."""
    if "type" not in data:
        raise ValueError("Geometry type not specified")
    kind = str(data["type"]).lower()
    scope = str(data.get("scope", "corner")).lower()
    if scope == "corner":
        _require_keys(data, {"type", "scope", "side", "name", "version", "units", "actuation", "spring",
                             "hardpoints", "config"}, "geometry")
        config = data["config"]
        return _build_corner(kind, str(data.get("name", "unnamed")), _side(data.get("side", "left")),
                             _hardpoints(data["hardpoints"]), config, data.get("actuation"), data.get("spring"),
                             vehicle=_vehicle(config, config.get("axle_position")), camber_shim=_shim(config, kind))
    if scope != "axle":
        raise ValueError(f"Unsupported geometry scope: '{scope}'")

    _require_keys(data, {"type", "scope", "name", "version", "units", "vehicle_config", "axle_config",
                         "hardpoints"}, "geometry")
    axle_config = data["axle_config"]
    left_shim = _shim(axle_config.get("left_setup") or {}, kind)
    if axle_config.get("right_setup") is not None:
        right_shim = _shim(axle_config["right_setup"], kind)
    else:  # mirrored right setup (build.py:310-318,357-375)
        right_shim = None if left_shim is None else left_shim.mirrored()
    shims = {Side.LEFT: left_shim, Side.RIGHT: right_shim}
    hp = data["hardpoints"]
    _require_keys(hp, {"left", "right", "center"}, "axle hardpoints")
    left = _hardpoints(hp["left"])
    right = _hardpoints(hp["right"]) if hp.get("right") is not None else _mirror(left)
    center = _hardpoints(hp.get("center") or {})
    arb = str(axle_config.get("anti_roll", {}).get("type", "none")).lower()
    heave = str(axle_config.get("heave_link", {}).get("type", "none")).lower()
    if arb not in ("none", "u_bar", "t_bar"):
        raise ValueError(f"Unsupported anti-roll type: {arb}")
    if heave not in ("none", "rocker_to_rocker"):
        raise ValueError(f"Unsupported heave-link type: {heave}")
    has_rocker = kind == "double_wishbone" and str(axle_config.get("actuation", {}).get("type", "")).lower() == "pushrod_rocker"
    strut = ", which a MacPherson corner does not provide" if kind == "macpherson" else ""
    if arb != "none" and not has_rocker:  # schema/geometry.py:128-137,196-208
        raise ValueError("The implemented anti-roll mechanism requires pushrod-rocker actuation" + strut)
    if heave != "none" and not has_rocker:
        raise ValueError("A rocker-to-rocker heave link requires pushrod-rocker actuation" + strut)
    external: tuple = ()
    droplinks: dict = {}
    if arb != "none":  # build.py:147-179: the droplink's rocker pickup belongs to the corner, its bar end to the axle
        external = (PointID.DROPLINK_ROCKER,)
        arm = PointID.DROPLINK_U_BAR if arb == "u_bar" else PointID.DROPLINK_T_BAR
        for side, points in ((Side.LEFT, left), (Side.RIGHT, right)):
            if arm not in points:
                raise ValueError(f"{side.name} {arb.replace('_', '-')} requires {arm.name}")
            droplinks[side] = points.pop(arm)
    elif center:
        raise ValueError("Axle without anti-roll hardware does not accept center points")
    if heave != "none":
        external = (*external, PointID.HEAVE_LINK_ROCKER)
    name = str(data.get("name", "unnamed"))
    corners = {}
    for side, points in ((Side.LEFT, left), (Side.RIGHT, right)):
        corners[side] = _build_corner(kind, f"{name}_{side.name.lower()}", side, points, axle_config,
                                      axle_config.get("actuation"), axle_config.get("spring"), external,
                                      vehicle=_vehicle(data["vehicle_config"] or {}, axle_config.get("axle_position")),
                                      camber_shim=shims[side])
    return AxleSuspension(name=name, corners=corners, arb_center_points=center, arb_droplink_points=droplinks,
                          arb_kind=arb if arb != "none" else "", heave_link=heave != "none")


def build_sweep(data: Mapping[str, Any], suspension: Suspension | None = None) -> SweepConfig:
    """Expand a sweep mapping (``schema/sweep.py:137-196``): linspace ranges, index-paired."""
    _require_keys(data, {"version", "steps", "targets"}, "sweep")
    if int(data.get("version", 1)) != 1:
        raise ValueError(f"Unsupported sweep version: {data.get('version')}")
    steps = data.get("steps")
    dimensions = []
    for spec in data["targets"]:
        _require_keys(spec, {"point", "direction", "name", "side", "mode", "start", "stop", "values"}, "sweep target")
        point = _point_id(spec["point"])
        if spec.get("values") is not None:
            values = [float(v) for v in spec["values"]]
        else:
            if spec.get("start") is None or spec.get("stop") is None:
                raise ValueError(f"Target '{spec.get('name') or point.name}': must specify either 'values' or both 'start' and 'stop'")
            if steps is None:
                raise ValueError(f"Target '{spec.get('name') or point.name}': no 'steps' count available (specify at target or file level)")
            values = list(np.linspace(float(spec["start"]), float(spec["stop"]), int(steps)))
        direction_spec = spec["direction"]
        _require_keys(direction_spec, {"axis", "vector"}, "direction")
        if ("axis" in direction_spec) == ("vector" in direction_spec):
            raise ValueError("Specify exactly one of 'axis' or 'vector'")
        if "axis" in direction_spec:
            direction = PointTargetAxis(_AXES[str(direction_spec["axis"]).lower()])
        else:
            vector = np.asarray(direction_spec["vector"], dtype=np.float64)
            if vector.shape != (3,):
                raise ValueError(f"Vector must be 3D, got shape {vector.shape}")
            norm = float(np.linalg.norm(vector))
            if norm == 0.0:
                raise ValueError("Direction vector cannot be zero")
            unit = vector / norm
            axis = next((a for a in Axis if np.allclose(unit, np.eye(3)[int(a)])), None)
            direction = PointTargetAxis(axis) if axis is not None else PointTargetVector(unit)
        side = _side(spec["side"]) if spec.get("side") is not None else None
        if side is Side.CENTER:
            raise ValueError("Sweep target side must be 'left' or 'right'.")
        if suspension is not None:
            key = suspension.resolve_target_key(point, side)
            state = suspension.initial_state()
            if key not in state.positions:
                raise ValueError(f"Sweep target point '{key.name}' is not present in this suspension.")
            if key not in state.free_points and key not in suspension.derived_spec().functions:
                raise ValueError(f"Sweep target point '{key.name}' is fixed in this suspension.")
        else:
            if side is not None:
                raise ValueError(f"Sweep target for '{point.name}' specifies a 'side', which requires a suspension context to resolve.")
            key = point
        mode = TargetPositionMode(str(spec.get("mode", "relative")).lower())
        dimensions.append([PointTarget(key, direction, value, mode) for value in values])
    lengths = {len(d) for d in dimensions}
    if len(lengths) > 1:
        raise ValueError(f"All targets must have the same length, got: {sorted(lengths)}")
    config = SweepConfig(dimensions)
    if suspension is not None:
        validate_sweep_controls(config, suspension.actuator_dofs())
    return config


def _read_yaml_mapping(path, kind: str) -> dict:
    import yaml

    path = Path(path)
    try:
        with open(path, "r", encoding="utf-8") as fh:
            data = yaml.safe_load(fh)
    except FileNotFoundError:
        raise FileNotFoundError(f"{kind} file not found: {path}")
    except yaml.YAMLError as error:
        raise ValueError(f"Error parsing {kind.lower()} file: {error}") from error
    if data is None:
        raise ValueError(f"{kind} file is empty: {path}")
    if not isinstance(data, dict):
        raise ValueError(f"{kind} file must contain a YAML mapping: {path}")
    return data


def load_geometry(path) -> Suspension:
    """``cli/io/loaders.py:29``."""
    return build_suspension(_read_yaml_mapping(path, "Geometry"))


def load_sweep(path, suspension: Suspension | None = None) -> SweepConfig:
    """``cli/io/sweep_loader.py:12``."""
    return build_sweep(_read_yaml_mapping(path, "Sweep"), suspension)
