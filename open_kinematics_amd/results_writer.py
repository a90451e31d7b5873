"""
Wide-form result files of a batched run, in the reference's file format.

The reference solves step by step and therefore collects one Python row per step
(``kinematics.cli.io.results_writer``); a batched device run has its results as whole arrays
already, so this module is columnar: a :class:`ResultTable` holds one NumPy array per column and
the two writers go from those arrays to the file in bulk (``pyarrow`` arrays without copies, CSV
text one column at a time).  A million-row ensemble never becomes a million Python objects.

File format (version 3, what the reference's readers expect):

* columns ``step_index, solver_converged, solver_max_residual, solver_nfev``, the metric columns in
  the caller's order, then ``<point>_x / _y / _z`` per output point under its public lower-case name
  (``core/export.py:11-27``, ``cli/io/results_writer.py:149-180``);
* CSV: ``# key: value`` provenance lines, ``# column_units: {json}``, ``#``, header, rows; cells are
  Python ``repr`` of the value, missing metric values are empty (``results_writer.py:369-460``);
* Parquet: provenance as JSON under schema key ``kinematics_meta``, the unit of a column in its
  field metadata ``unit`` (``results_writer.py:225-366``).

``SolutionFrame`` / ``CsvWriter`` / ``ParquetWriter`` keep the reference's per-step calling
convention (``add_frame`` then ``write``) on top of the same table for code written against it,
including ``SolutionFrame.metric_specs`` (``results_writer.py:88-103``; any object with a
``unit.symbol``, e.g. this module's :class:`MetricSpec`).
"""

from __future__ import annotations

import hashlib
import io
import json
import time
from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path
from typing import Any, Iterable, Mapping, Sequence

import numpy as np

from .solver import SolverInfo

FORMAT_VERSION = "3"
METADATA_KEY = b"kinematics_meta"
STANDARD_COLUMNS = ("step_index", "solver_converged", "solver_max_residual", "solver_nfev")


# ---- units ---------------------------------------------------------------------------------------------------------


class MetricUnit(Enum):
    """Scalar units of the metric catalog (``core/metrics/units.py:9-25``)."""

    MM = "mm"
    DEG = "deg"
    PERCENT = "%"

    @property
    def symbol(self) -> str:
        return self.value

    def __truediv__(self, other):
        return MetricUnitQuotient(self, other) if isinstance(other, MetricUnit) else NotImplemented


@dataclass(frozen=True)
class MetricUnitQuotient:
    """Unit of a derivative column (``core/metrics/units.py:28-42``)."""

    numerator: MetricUnit
    denominator: MetricUnit

    @property
    def symbol(self) -> str:
        return f"{self.numerator.symbol}/{self.denominator.symbol}"

    def __str__(self) -> str:
        return self.symbol


class MetricKind(str, Enum):
    STATE = "state"
    DERIVATIVE = "derivative"


class Scope(str, Enum):
    CORNER = "corner"
    AXLE = "axle"


@dataclass(frozen=True)
class MetricSpec:
    """Identity of a metric column as the reference declares it (``core/metrics/registry.py:36-44``)."""

    key: str
    label: str
    unit: Any
    kind: MetricKind = MetricKind.STATE
    scope: Scope = Scope.CORNER
    component: str | None = None


_MM, _DEG, _PCT = "mm", "deg", "%"
# units of the metric columns this package produces (reference: metrics/catalog.py, axle_metrics.py, mechanisms.py)
METRIC_UNITS = {
    **{k: _DEG for k in ("camber", "caster", "kpi", "roadwheel_angle", "svsa_angle", "roll", "rocker_angle", "torsion_bar_twist",
                         "arb_arm_angle", "arb_twist", "t_bar_heave_angle")},
    **{k: _MM for k in ("wheel_travel", "half_track", "scrub_radius", "mechanical_trail", "svic_x", "svic_z", "svsa_length",
                        "fvic_y", "fvic_z", "fvsa_length", "damper_length", "heave", "ride_height_change", "track",
                        "roll_center_y", "roll_center_z", "rack_displacement", "t_bar_center_x", "heave_link_length")},
    **{k: _PCT for k in ("anti_dive", "anti_lift", "anti_squat")},
}


def metric_unit(column: str) -> str | None:
    """
    Unit symbol of a metric column by name: the catalog's unit; ``deriv_<response>_wrt_<driver>`` is the response's unit
    per millimetre (every driver the reference declares is a length, ``metrics/units.py:27-42``); a ``_left`` /
    ``_right`` suffix does not change the unit.
    """
    if column in METRIC_UNITS:
        return METRIC_UNITS[column]
    if column.startswith("deriv_") and "_wrt_" in column:
        response = column[len("deriv_"):].split("_wrt_")[0]
        unit = METRIC_UNITS.get(response, _MM if response.endswith(("_x", "_y", "_z")) else None)
        return None if unit is None else f"{unit}/mm"
    for suffix in ("_left", "_right"):
        if column.endswith(suffix):
            return metric_unit(column[: -len(suffix)])
    return None


def _unit_symbol(spec_or_unit) -> str | None:
    """A unit string from a spec object (``.unit.symbol``), a unit object (``.symbol``) or a plain string."""
    if spec_or_unit is None:
        return None
    unit = getattr(spec_or_unit, "unit", spec_or_unit)
    symbol = getattr(unit, "symbol", unit)
    return None if symbol is None else str(symbol)


# ---- names ---------------------------------------------------------------------------------------------------------


def point_key_name(key) -> str:
    """Public name of a point key (``primitives/point_ref.py:92-94``; side-qualified keys carry theirs)."""
    name = getattr(key, "lower_name", None)
    return name if name is not None else key.name.lower()


def flatten_positions(positions: Mapping, output_points: Sequence) -> dict:
    """``core/export.py:11-27``: the selected points of a state under their public names, as xyz tuples."""
    flat = {}
    for point in output_points:
        value = positions.get(point)
        if value is not None:
            x, y, z = np.asarray(getattr(value, "data", value), dtype=np.float64)[:3]
            flat[point_key_name(point)] = (float(x), float(y), float(z))
    return flat


def compute_file_hash(path) -> str:
    """SHA-256 of a provenance file, empty when it cannot be read (``results_writer.py:69-83``)."""
    try:
        digest = hashlib.sha256()
        with open(path, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 20), b""):
                digest.update(chunk)
        return digest.hexdigest()
    except OSError:
        return ""


def provenance(geometry_path=None, sweep_path=None, **extra: str) -> dict:
    """The metadata block of a result file, in the reference's key order."""
    meta = {"format_version": FORMAT_VERSION, "timestamp": str(time.time()), **extra}
    for label, path in (("geometry", geometry_path), ("sweep", sweep_path)):
        if path is not None:
            meta[f"{label}_path"] = str(path)
            meta[f"{label}_hash"] = compute_file_hash(path)
    return meta


# ---- the table -----------------------------------------------------------------------------------------------------


def _host(values) -> np.ndarray:
    return np.asarray(values.detach().cpu() if hasattr(values, "detach") else values)


def _coordinate_columns(positions) -> np.ndarray:
    """``[B, n_out, 3]`` -> ``[n_out * 3, B]`` with every coordinate column contiguous: a device tensor is transposed where
    it lives, a host array in row blocks that stay in cache (a strided gather per column would walk the array 3 n_out times)."""
    if hasattr(positions, "detach"):
        flat = positions.detach().reshape(positions.shape[0], -1)
        return flat.t().contiguous().cpu().numpy().astype(np.float64, copy=False)
    flat = np.asarray(positions, dtype=np.float64).reshape(len(positions), -1)
    out = np.empty((flat.shape[1], flat.shape[0]), dtype=np.float64)
    for lo in range(0, flat.shape[0], 4096):
        out[:, lo:lo + 4096] = flat[lo:lo + 4096].T
    return out


class ResultTable:
    """
    Columns of a result file: ``name -> 1-D array`` in file order (bool, int64, float64 or str), a unit per column where
    one is known, and per column an optional mask of MISSING values (the reference's ``None``: an empty CSV cell, a Parquet
    null).  A NaN that is not masked is a value and is written as one (``nan`` / NaN), like the reference writes it.
    """

    def __init__(self) -> None:
        self.columns: dict[str, np.ndarray] = {}
        self.units: dict[str, str] = {}
        self.missing: dict[str, np.ndarray] = {}
        self.n_rows: int | None = None

    def add(self, name: str, values, unit: str | None = None, missing=None) -> None:
        array = _host(values)
        if array.ndim != 1:
            raise ValueError(f"column '{name}' is not one-dimensional: shape {array.shape}")
        if self.n_rows is None:
            self.n_rows = int(array.shape[0])
        elif array.shape[0] != self.n_rows:
            raise ValueError(f"column '{name}' has {array.shape[0]} rows, the table has {self.n_rows}")
        if name in self.columns:
            raise ValueError(f"duplicate column '{name}'")
        if array.dtype == np.bool_:
            pass
        elif np.issubdtype(array.dtype, np.integer):
            array = array.astype(np.int64, copy=False)
        elif np.issubdtype(array.dtype, np.floating):
            array = array.astype(np.float64, copy=False)
        elif array.dtype.kind not in "UO":
            raise ValueError(f"column '{name}' has unsupported dtype {array.dtype}")
        self.columns[name] = array
        if missing is not None:
            mask = np.asarray(missing, dtype=np.bool_)
            if mask.shape != array.shape:
                raise ValueError(f"column '{name}': the missing-value mask has shape {mask.shape}, the column {array.shape}")
            if mask.any():
                self.missing[name] = mask
        if unit is not None:
            self.set_unit(name, unit)

    def set_unit(self, name: str, unit: str) -> None:
        known = self.units.get(name)
        if known is not None and known != unit:
            raise ValueError(f"Conflicting units for column '{name}': {known} and {unit}")
        self.units[name] = unit

    def sorted_by_step(self) -> "ResultTable":
        """Rows in ``step_index`` order (stable); the table itself when they already are."""
        steps = self.columns.get("step_index")
        if steps is None or self.n_rows in (None, 0) or np.all(steps[1:] >= steps[:-1]):
            return self
        order = np.argsort(steps, kind="stable")
        out = ResultTable()
        out.n_rows, out.units = self.n_rows, dict(self.units)
        out.columns = {name: values[order] for name, values in self.columns.items()}
        out.missing = {name: mask[order] for name, mask in self.missing.items()}
        return out

    @classmethod
    def from_batch(cls, program, positions, info, metrics: Mapping[str, Any] | None = None,
                   metric_units: Mapping[str, Any] | None = None, step_index=None) -> "ResultTable":
        """
        The table of a batched device run: ``positions [B, n_out, 3]`` (tensor or array), ``info`` the structured array of
        ``BatchResult.info()``, ``metrics`` ``name -> [B]`` in file order (NaN = undefined: the device's spelling of the
        reference's ``None``, written as missing), ``metric_units`` overrides / additions to the units known by name
        (strings, units or specs).  A NaN in a solver column or a coordinate is a value and stays one.
        """
        from ._abi import INFO_CONVERGED, INFO_FAILED, INFO_RESIDUAL_EXCEEDED

        shape = tuple(positions.shape)
        if len(shape) != 3 or shape[2] != 3 or shape[1] != len(program.out_point):
            raise ValueError(f"positions must be [B, {len(program.out_point)}, 3], got {shape}")
        table = cls()
        flags = np.asarray(info["flags"])
        table.add("step_index", np.arange(shape[0], dtype=np.int64) if step_index is None else step_index)
        table.add("solver_converged", ((flags & INFO_CONVERGED) != 0) & ((flags & (INFO_RESIDUAL_EXCEEDED | INFO_FAILED)) == 0))
        table.add("solver_max_residual", np.asarray(info["max_residual"], dtype=np.float64))
        table.add("solver_nfev", np.asarray(info["nfev"]))
        overrides = {name: _unit_symbol(u) for name, u in (metric_units or {}).items()}
        for name, values in (metrics or {}).items():
            column = np.asarray(_host(values), dtype=np.float64)
            table.add(name, column, overrides.get(name) or metric_unit(name), missing=np.isnan(column))
        coordinates = _coordinate_columns(positions)
        for k, index in enumerate(program.out_point):
            name = point_key_name(program.point_keys[index])
            for axis, letter in enumerate("xyz"):
                table.add(f"{name}_{letter}", coordinates[3 * k + axis], _MM)
        return table

    # -- writers --

    def write_parquet(self, path, metadata: Mapping[str, str]) -> None:
        import pyarrow as pa
        import pyarrow.parquet as pq

        table = self.sorted_by_step()
        arrays, fields = [], []
        for name, values in table.columns.items():
            missing = table.missing.get(name)
            if values.dtype.kind in "UO":
                array = pa.array([None if v is None else str(v) for v in values.tolist()], type=pa.string())
            else:
                array = pa.array(values, mask=missing)
            unit = table.units.get(name)
            arrays.append(array)
            fields.append(pa.field(name, array.type, metadata={b"unit": unit.encode("utf-8")} if unit else None))
        schema = pa.schema(fields, metadata={METADATA_KEY: json.dumps(dict(metadata)).encode("utf-8")})
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        # (dictionary encoding only where values repeat: on float columns it costs 6x the write time and gains nothing)
        repeating = [f.name for f in fields if not pa.types.is_floating(f.type)]
        pq.write_table(pa.Table.from_arrays(arrays, schema=schema), path, use_dictionary=repeating)

    def write_csv(self, path, metadata: Mapping[str, str]) -> None:
        table = self.sorted_by_step()
        text = io.StringIO()
        for key, value in metadata.items():
            text.write(f"# {key}: {value}\n")
        text.write(f"# column_units: {json.dumps(table.units, sort_keys=True)}\n#\n")
        text.write(",".join(_csv_cell(name) for name in table.columns) + "\n")
        cells = [_csv_column(values, table.missing.get(name)) for name, values in table.columns.items()]
        text.write("\n".join(map(",".join, zip(*cells))))
        if table.n_rows:
            text.write("\n")
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(text.getvalue(), newline="")


def _csv_cell(value: str) -> str:
    """Minimal quoting, as Python's ``csv`` module does it."""
    if any(ch in value for ch in ',"\r\n'):
        return '"' + value.replace('"', '""') + '"'
    return value


def _csv_column(values: np.ndarray, missing=None) -> list[str]:
    """Cells of one column: ``repr`` of bools / ints / floats (what ``csv.writer`` emits; a NaN reads ``nan``), empty where
    the value is missing (the reference's ``None``)."""
    if values.dtype.kind in "UO":
        return ["" if v is None else _csv_cell(str(v)) for v in values.tolist()]
    cells = [repr(v) for v in values.tolist()]
    if missing is not None:
        for k in np.nonzero(missing)[0].tolist():
            cells[k] = ""
    return cells


# ---- the reference's per-step calling convention ---------------------------------------------------------------------


@dataclass
class SolutionFrame:
    """
    One step's results as the reference's writer takes them (``results_writer.py:88-103``).  ``metric_specs`` maps metric
    names to spec objects (``.unit.symbol``); ``metric_units`` (this package's earlier spelling) maps them to unit strings.
    """

    positions: dict
    solver_info: SolverInfo
    metrics: dict = field(default_factory=dict)
    metric_specs: dict = field(default_factory=dict)
    metric_units: dict = field(default_factory=dict)


class _FrameWriter:
    """
    Collects frames column-wise: per column one Python list that grows by a value per frame; ``write`` turns the lists into
    a :class:`ResultTable`.  The column set is fixed by the first frame.
    """

    def __init__(self, output_path, geometry_path=None, sweep_path=None, **extra_metadata: str):
        self.output_path = Path(output_path)
        self.metadata = provenance(geometry_path, sweep_path, **extra_metadata)
        self.column_units: dict[str, str] = {}
        self._names: tuple[str, ...] | None = None
        self._values: list[list] = []
        self._n = 0
        self._column_error: str | None = None  # a frame whose columns differ from the first one's: reported by write(), as the reference does

    @property
    def frames(self) -> int:
        return self._n

    def _unit(self, column: str, unit: str) -> None:
        known = self.column_units.get(column)
        if known is not None and known != unit:
            raise ValueError(f"Conflicting units for column '{column}': {known} and {unit}")
        self.column_units[column] = unit

    def add_frame(self, frame_index: int, frame: SolutionFrame) -> None:
        info = frame.solver_info
        names = list(STANDARD_COLUMNS)
        row: list = [int(frame_index), bool(info.converged), float(info.max_residual), int(info.nfev)]
        for name, value in frame.metrics.items():
            if isinstance(value, (list, tuple, np.ndarray)):
                raise ValueError(f"Frame {self._n}, column '{name}' contains nested data: {value!r}. Expected scalar value.")
            names.append(name)
            row.append(value)
            unit = _unit_symbol(frame.metric_specs.get(name)) or frame.metric_units.get(name) or metric_unit(name)
            if unit is not None:
                self._unit(name, unit)
        for point, xyz in frame.positions.items():
            for letter, coordinate in zip("xyz", xyz):
                names.append(f"{point}_{letter}")
                row.append(float(coordinate))
                self._unit(names[-1], _MM)
        if self._names is None:
            self._names, self._values = tuple(names), [[] for _ in names]
        elif tuple(names) != self._names:
            want, got = set(self._names), set(names)
            if want == got:  # same set in another order: file the values under their names
                by_name = dict(zip(names, row))
                row = [by_name[n] for n in self._names]
            else:
                parts = ([f"Missing columns: {sorted(want - got)}"] if want - got else []) + \
                        ([f"Extra columns: {sorted(got - want)}"] if got - want else [])
                if self._column_error is None:
                    self._column_error = f"Frame {self._n} has inconsistent columns - {', '.join(parts)}"
                self._n += 1
                return
        for column, value in zip(self._values, row):
            column.append(value)
        self._n += 1

    def table(self) -> ResultTable:
        if self._n == 0:
            raise ValueError("No frames to write")
        if self._column_error is not None:
            raise ValueError(self._column_error)
        table = ResultTable()
        for name, values in zip(self._names, self._values):
            # the reference's type inference (cli/io/results_writer.py:316-335), None = a missing value
            bad = next((v for v in values if v is not None and not isinstance(v, (bool, int, float, str))), None)
            if bad is not None:
                raise ValueError(f"column '{name}' contains unexpected type {type(bad).__name__}: {bad!r}. "
                                 "Expected bool, int, float, str, or None.")
            missing = np.fromiter((v is None for v in values), dtype=np.bool_, count=len(values))
            if all(isinstance(v, bool) or v is None for v in values):  # (a column of nothing but None is a bool column of nulls)
                array = np.asarray([bool(v) for v in values], dtype=np.bool_)
            elif all(isinstance(v, int) or v is None for v in values) and not name.endswith(("_x", "_y", "_z")):
                array = np.asarray([0 if v is None else v for v in values], dtype=np.int64)
            elif all(isinstance(v, (int, float)) or v is None for v in values):
                array = np.asarray([np.nan if v is None else float(v) for v in values], dtype=np.float64)
            else:  # strings, or strings mixed with numbers: a string column
                array = np.asarray([None if v is None else str(v) for v in values], dtype=object)
                missing = None
            table.add(name, array, self.column_units.get(name), missing=missing)
        return table


class CsvWriter(_FrameWriter):
    def write(self) -> None:
        self.table().write_csv(self.output_path, self.metadata)


class ParquetWriter(_FrameWriter):
    def write(self) -> None:
        self.table().write_parquet(self.output_path, self.metadata)


def write_batch(path, program, positions, info, metrics: Mapping[str, Any] | None = None,
                metric_units: Mapping[str, Any] | None = None, geometry_path=None, sweep_path=None, **extra_metadata: str) -> ResultTable:
    """A batched run straight to a ``.csv`` or ``.parquet`` file (by suffix); returns the table that was written."""
    table = ResultTable.from_batch(program, positions, info, metrics, metric_units)
    meta = provenance(geometry_path, sweep_path, **extra_metadata)
    if str(path).lower().endswith(".csv"):
        table.write_csv(path, meta)
    else:
        table.write_parquet(path, meta)
    return table


def frames_from_batch(program, positions, info, metrics: Mapping[str, Any] | None = None,
                      metric_units: Mapping[str, str] | None = None) -> Iterable[SolutionFrame]:
    """
    The rows of a batched run as frames, for code that feeds a per-step writer (small runs: ``write_batch`` /
    ``ResultTable.from_batch`` are the bulk path).  NaN metric values become ``None`` like the reference's undefined ones.
    """
    table = ResultTable.from_batch(program, positions, info, metrics, metric_units)
    names = list(table.columns)
    n_metric = len(metrics or {})
    metric_names = names[4:4 + n_metric]
    point_names = [n[:-2] for n in names[4 + n_metric::3]]
    rows = zip(*(table.columns[n].tolist() for n in names))
    units = {n: table.units[n] for n in metric_names if n in table.units}
    for row in rows:
        values = [None if isinstance(v, float) and v != v else v for v in row[4:4 + n_metric]]
        coords = row[4 + n_metric:]
        yield SolutionFrame(
            positions={p: tuple(coords[3 * k:3 * k + 3]) for k, p in enumerate(point_names)},
            solver_info=SolverInfo(converged=bool(row[1]), nfev=int(row[3]), max_residual=float(row[2])),
            metrics=dict(zip(metric_names, values)), metric_units=units)


def frames_from_states(states, solver_infos, metric_rows, output_points) -> list[SolutionFrame]:
    """
    Frames of a solved sweep in object form (``solve_sweep`` -> ``compute_sweep_metrics`` -> writer): ``metric_rows`` are
    the rows ``compute_sweep_metrics`` returns (mappings, or objects with ``flat_row()``).
    """
    frames = []
    for state, info, row in zip(states, solver_infos, metric_rows):
        flat = dict(row.flat_row() if hasattr(row, "flat_row") else row)
        units = {name: unit for name in flat if (unit := metric_unit(name)) is not None}
        frames.append(SolutionFrame(flatten_positions(state.positions, output_points), info, flat, metric_units=units))
    return frames
