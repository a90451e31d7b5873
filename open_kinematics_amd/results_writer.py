"""
Wide-form result files in the reference's format: drop-in for
``kinematics.cli.io.results_writer`` (``SolutionFrame``, ``CsvWriter``, ``ParquetWriter``,
``results_writer.py:63-460``) and ``kinematics.core.export.flatten_positions``
(``core/export.py:11-27``), plus ``frames_from_batch`` which turns the tensors of a batched device
run (positions, info records, metric columns) into frames without per-step Python objects upstream.

Format (version 3): columns ``step_index, solver_converged, solver_max_residual, solver_nfev``, then
the metric columns in the order given, then ``<point>_x/_y/_z`` for every output point with the
point's lower snake-case name; CSV files carry the metadata as ``# key: value`` comment lines
followed by ``# column_units: {json}`` and ``#``; Parquet files carry it under the schema metadata key
``kinematics_meta`` and per-field ``unit`` metadata.
"""

from __future__ import annotations

import csv
import hashlib
import json
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Mapping, Sequence

import numpy as np

from .solver import SolverInfo

FORMAT_VERSION = "3"
METADATA_KEY = b"kinematics_meta"
STANDARD_COLUMNS = ("step_index", "solver_converged", "solver_max_residual", "solver_nfev")

# units of the metric columns this package produces (reference: metrics/catalog.py, metrics/units.py)
METRIC_UNITS = {
    "camber": "deg", "caster": "deg", "kpi": "deg", "roadwheel_angle": "deg",
    "wheel_travel": "mm", "half_track": "mm", "scrub_radius": "mm", "mechanical_trail": "mm",
    "svic_x": "mm", "svic_z": "mm", "svsa_length": "mm", "fvic_y": "mm", "fvic_z": "mm", "fvsa_length": "mm",
    "damper_length": "mm", "svsa_angle": "deg", "anti_dive": "%", "anti_lift": "%", "anti_squat": "%",
    # axle scope (metrics/axle_metrics.py, metrics/registry.py)
    "heave": "mm", "roll": "deg", "ride_height_change": "mm", "track": "mm", "roll_center_y": "mm",
    "roll_center_z": "mm", "rack_displacement": "mm",
    # topology-specific rotations (corner/mechanisms.py:58-74, axle/mechanisms.py:76-91)
    "rocker_angle": "deg", "torsion_bar_twist": "deg", "arb_arm_angle": "deg", "arb_twist": "deg",
}


def point_key_name(key) -> str:
    """``primitives/point_ref.py:92-94`` (side-qualified keys already carry their public name)."""
    name = getattr(key, "lower_name", None)
    return name if name is not None else key.name.lower()


def flatten_positions(positions: Mapping, output_points: Sequence) -> dict:
    """``core/export.py:11-27``: selected typed positions -> public names and xyz tuples."""
    flattened = {}
    for point in output_points:
        position = positions.get(point)
        if position is None:
            continue
        raw = np.asarray(getattr(position, "data", position), dtype=np.float64)
        flattened[point_key_name(point)] = (float(raw[0]), float(raw[1]), float(raw[2]))
    return flattened


def compute_file_hash(path) -> str:
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()
    except Exception:
        return ""


@dataclass
class SolutionFrame:
    """``results_writer.py:88-103``; ``metric_units`` replaces the reference's ``metric_specs``."""

    positions: dict
    solver_info: SolverInfo
    metrics: dict = field(default_factory=dict)
    metric_units: dict = field(default_factory=dict)


class BaseResultsWriter:
    """``results_writer.py:106-222``."""

    def __init__(self, output_path, geometry_path=None, sweep_path=None, **extra_metadata: str):
        self.output_path = Path(output_path)
        self.frames: list[dict[str, Any]] = []
        self.column_units: dict[str, str] = {}
        self.metadata: dict[str, str] = {"format_version": FORMAT_VERSION, "timestamp": str(time.time()), **extra_metadata}
        if geometry_path is not None:
            self.metadata["geometry_path"] = str(geometry_path)
            self.metadata["geometry_hash"] = compute_file_hash(geometry_path)
        if sweep_path is not None:
            self.metadata["sweep_path"] = str(sweep_path)
            self.metadata["sweep_hash"] = compute_file_hash(sweep_path)

    def add_frame(self, frame_index: int, frame: SolutionFrame) -> None:
        row: dict[str, Any] = {"step_index": int(frame_index)}
        row["solver_converged"] = bool(frame.solver_info.converged)
        row["solver_max_residual"] = float(frame.solver_info.max_residual)
        row["solver_nfev"] = int(frame.solver_info.nfev)
        for name, value in frame.metrics.items():
            row[name] = value
            unit = frame.metric_units.get(name, METRIC_UNITS.get(name))
            if unit is not None:
                self._record_column_unit(name, unit)
        for point_id, (x, y, z) in frame.positions.items():
            row[f"{point_id}_x"], row[f"{point_id}_y"], row[f"{point_id}_z"] = float(x), float(y), float(z)
            for axis in ("x", "y", "z"):
                self._record_column_unit(f"{point_id}_{axis}", "mm")
        self.frames.append(row)

    def _record_column_unit(self, column: str, unit: str) -> None:
        existing = self.column_units.get(column)
        if existing is not None and existing != unit:
            raise ValueError(f"Conflicting units for column '{column}': {existing} and {unit}")
        self.column_units[column] = unit

    def build_column_list(self) -> list[str]:
        if not self.frames:
            raise ValueError("No frames to validate")
        columns = list(self.frames[0].keys())
        first = set(columns)
        for i, frame in enumerate(self.frames[1:], 1):
            got = set(frame.keys())
            if got != first:
                parts = []
                if first - got:
                    parts.append(f"Missing columns: {sorted(first - got)}")
                if got - first:
                    parts.append(f"Extra columns: {sorted(got - first)}")
                raise ValueError(f"Frame {i} has inconsistent columns - {', '.join(parts)}")
        return columns

    def _validated_columns(self) -> list[str]:
        if not self.frames:
            raise ValueError("No frames to write")
        self.frames.sort(key=lambda r: r["step_index"])
        columns = self.build_column_list()
        for index, frame in enumerate(self.frames):
            for col in columns:
                val = frame.get(col)
                if val is None:
                    continue
                if isinstance(val, (list, tuple, np.ndarray)):
                    raise ValueError(f"Frame {index}, column '{col}' contains nested data: {val!r}. "
                                     "Expected scalar value. Check position data flattening.")
                if not isinstance(val, (bool, int, float, str)):
                    raise ValueError(f"Frame {index}, column '{col}' contains unexpected type "
                                     f"{type(val).__name__}: {val!r}. Expected bool, int, float, str, or None.")
        return columns

    def write(self) -> None:
        raise NotImplementedError


class CsvWriter(BaseResultsWriter):
    """``results_writer.py:369-460``."""

    def write(self) -> None:
        columns = self._validated_columns()
        self.output_path.parent.mkdir(parents=True, exist_ok=True)
        with open(self.output_path, "w", newline="") as fh:
            for key, value in self.metadata.items():
                fh.write(f"# {key}: {value}\n")
            fh.write(f"# column_units: {json.dumps(self.column_units, sort_keys=True)}\n")
            fh.write("#\n")
            writer = csv.DictWriter(fh, fieldnames=columns, lineterminator="\n")
            writer.writeheader()
            for frame in self.frames:
                writer.writerow({col: frame.get(col) for col in columns})


class ParquetWriter(BaseResultsWriter):
    """``results_writer.py:225-366``."""

    def write(self) -> None:
        import pyarrow as pa
        import pyarrow.parquet as pq

        columns = self._validated_columns()
        arrays, fields = [], []
        for col in columns:
            values = [frame.get(col) for frame in self.frames]
            if all(isinstance(v, bool) or v is None for v in values):
                arr = pa.array(values, type=pa.bool_())
            elif all(isinstance(v, int) or v is None for v in values) and not col.endswith(("_x", "_y", "_z")):
                arr = pa.array(values, type=pa.int64())
            elif all(isinstance(v, (int, float)) or v is None for v in values):
                arr = pa.array([None if v is None else float(v) for v in values], type=pa.float64())
            else:
                arr = pa.array([None if v is None else str(v) for v in values], type=pa.string())
            arrays.append(arr)
            unit = self.column_units.get(col)
            fields.append(pa.field(col, arr.type, metadata={b"unit": unit.encode("utf-8")} if unit else None))
        table = pa.Table.from_arrays(arrays, schema=pa.schema(fields))
        table = table.replace_schema_metadata({**(table.schema.metadata or {}),
                                               METADATA_KEY: json.dumps(self.metadata).encode("utf-8")})
        self.output_path.parent.mkdir(parents=True, exist_ok=True)
        pq.write_table(table, self.output_path)


def frames_from_batch(program, positions, info, metrics: Mapping[str, Any] | None = None,
                      metric_units: Mapping[str, str] | None = None) -> list[SolutionFrame]:
    """
    Frames of a batched device run: ``positions [B, n_out, 3]`` and ``info`` (structured array of
    ``BatchResult.info()``) plus optional metric columns ``name -> [B]`` (NaN -> empty cell, like the
    reference's ``None`` for undefined geometry).  Column order of the metrics is the mapping's order.
    """
    from ._abi import INFO_CONVERGED, INFO_FAILED, INFO_RESIDUAL_EXCEEDED

    pos = np.asarray(positions.cpu() if hasattr(positions, "cpu") else positions, dtype=np.float64)
    names = [point_key_name(program.point_keys[k]) for k in program.out_point]
    columns = {}
    for name, values in (metrics or {}).items():
        columns[name] = np.asarray(values.cpu() if hasattr(values, "cpu") else values, dtype=np.float64)
    units = dict(metric_units or {})
    frames = []
    for b in range(pos.shape[0]):
        flags = int(info["flags"][b])
        ok = bool(flags & INFO_CONVERGED) and not flags & (INFO_RESIDUAL_EXCEEDED | INFO_FAILED)
        row_metrics = {}
        for name, values in columns.items():
            v = float(values[b])
            row_metrics[name] = None if v != v else v
        frames.append(SolutionFrame(
            positions={n: tuple(float(c) for c in pos[b, k]) for k, n in enumerate(names)},
            solver_info=SolverInfo(converged=ok, nfev=int(info["nfev"][b]), max_residual=float(info["max_residual"][b])),
            metrics=row_metrics, metric_units=units,
        ))
    return frames


def metric_unit(column: str) -> str | None:
    """
    Unit symbol of a metric column: the catalog's units, and for a derivative column
    ``deriv_<response>_wrt_<driver>`` the quotient ``<response unit>/<driver unit>`` (``metrics/units.py:27-42``;
    both drivers the reference declares — ``hub_z[_side]``, ``rack_displacement`` — are millimetres).
    """
    if column in METRIC_UNITS:
        return METRIC_UNITS[column]
    if column.startswith("deriv_") and "_wrt_" in column:
        response = column[len("deriv_"):].split("_wrt_")[0]
        unit = METRIC_UNITS.get(response, "mm" if response.endswith(("_x", "_y", "_z")) else None)
        return None if unit is None else f"{unit}/mm"
    for suffix in ("_left", "_right"):
        if column.endswith(suffix):
            return metric_unit(column[: -len(suffix)])
    return None


def frames_from_states(states, solver_infos, metric_rows, output_points) -> list[SolutionFrame]:
    """
    Frames of a solved sweep in the reference's object form (``cli`` path: ``solve_sweep`` ->
    ``compute_sweep_metrics`` -> writer): ``states`` / ``solver_infos`` as returned by ``solve_sweep``,
    ``metric_rows`` the rows of ``compute_sweep_metrics`` (``OrderedDict`` or ``AxleMetricRows``).
    """
    frames = []
    for state, info, row in zip(states, solver_infos, metric_rows):
        flat = row.flat_row() if hasattr(row, "flat_row") else row
        frames.append(SolutionFrame(
            positions=flatten_positions(state.positions, output_points), solver_info=info, metrics=dict(flat),
            metric_units={name: unit for name in flat if (unit := metric_unit(name)) is not None},
        ))
    return frames
