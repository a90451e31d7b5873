"""Wide-form result files in the reference's format (results_writer.py / export.py), CPU only."""

import csv
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from open_kinematics_amd.results_writer import (CsvWriter, ParquetWriter, SolutionFrame, flatten_positions,
                                                frames_from_batch)
from open_kinematics_amd.solver import SolverInfo


def _golden_csv():
    with open(os.path.join(GOLDEN, "e2e_output.csv"), encoding="utf-8") as fh:
        lines = fh.read().splitlines()
    meta = [ln for ln in lines if ln.startswith("#")]
    rows = list(csv.DictReader(ln for ln in lines if not ln.startswith("#")))
    return meta, rows


def test_csv_layout_matches_the_reference_golden(golden, tmp_path):
    """Same metadata comment block, same column order for the shared columns, same cell formatting."""
    arrays, program = golden("e2e_sweep")
    meta, rows = _golden_csv()
    pos = arrays["ref_default_pos"]
    info = np.zeros(pos.shape[0], dtype=[("flags", "<i4"), ("nfev", "<i4"), ("max_residual", "<f8")])
    info["flags"], info["nfev"], info["max_residual"] = 1, arrays["ref_default_nfev"], arrays["ref_default_maxres"]
    mg = dict(np.load(os.path.join(GOLDEN, "metrics_e2e_sweep.npz"), allow_pickle=False))
    names = ("camber", "caster", "kpi", "scrub_radius", "mechanical_trail", "roadwheel_angle", "wheel_travel", "half_track")
    order = ("camber", "caster", "kpi", "roadwheel_angle", "wheel_travel", "half_track", "scrub_radius", "mechanical_trail")
    metrics = {n: mg["values"][:, order.index(n)] for n in names}
    metrics["damper_length"] = np.full(pos.shape[0], np.nan)  # undefined for this topology: empty cells
    out = tmp_path / "out.csv"
    writer = CsvWriter(out, geometry_path=os.path.join(GOLDEN, "geometry", "geometry.yaml"),
                       sweep_path=os.path.join(GOLDEN, "geometry", "sweep.yaml"))
    for b, frame in enumerate(frames_from_batch(program, pos, info, metrics, {"damper_length": "mm"})):
        writer.add_frame(b, frame)
    writer.write()
    lines = out.read_text().splitlines()
    mine_meta = [ln for ln in lines if ln.startswith("#")]
    assert [ln.split(":")[0] for ln in mine_meta] == [ln.split(":")[0] for ln in meta]
    assert mine_meta[0] == "# format_version: 3" and mine_meta[-1] == "#"
    # the reference's fixture files: same content hash as the golden's provenance block
    assert mine_meta[3] == meta[3] and mine_meta[5] == meta[5]
    units = json.loads(mine_meta[-2].split(": ", 1)[1])
    ref_units = json.loads(meta[-2].split(": ", 1)[1])
    assert all(ref_units[k] == v for k, v in units.items())
    mine = list(csv.DictReader(ln for ln in lines if not ln.startswith("#")))
    ref_cols, my_cols = list(rows[0].keys()), list(mine[0].keys())
    assert my_cols[:4] == ref_cols[:4] == ["step_index", "solver_converged", "solver_max_residual", "solver_nfev"]
    assert [c for c in ref_cols if c in my_cols] == my_cols  # same relative order, subset of the metric columns
    point_names = {program.point_keys[k].lower_name for k in program.out_point}
    position_cols = [c for c in ref_cols if c[:-2] in point_names and c.endswith(("_x", "_y", "_z"))]
    assert len(position_cols) == 3 * program.n_out
    assert my_cols[-len(position_cols):] == position_cols == ref_cols[-len(position_cols):]
    assert len(mine) == len(rows)
    for a, b in zip(mine, rows):
        assert a["step_index"] == b["step_index"] and a["solver_converged"] == b["solver_converged"] == "True"
        assert a["damper_length"] == b["damper_length"] == ""
        for c in position_cols:
            assert abs(float(a[c]) - float(b[c])) <= 5e-5  # other platform
        for c in names:
            assert abs(float(a[c]) - float(b[c])) <= 5e-5


def test_parquet_carries_metadata_and_units(tmp_path):
    import pyarrow.parquet as pq

    out = tmp_path / "out.parquet"
    writer = ParquetWriter(out, tool="unit-test")
    for k in (1, 0):
        writer.add_frame(k, SolutionFrame({"wheel_center": (1.0, 2.0, 3.0 + k)}, SolverInfo(True, 5 + k, 1e-7),
                                          {"camber": -1.5, "damper_length": None}, {"damper_length": "mm"}))
    writer.write()
    table = pq.read_table(out)
    meta = json.loads(table.schema.metadata[b"kinematics_meta"])
    assert meta["format_version"] == "3" and meta["tool"] == "unit-test"
    assert table.column("step_index").to_pylist() == [0, 1]  # sorted by step
    assert table.schema.field("camber").metadata[b"unit"] == b"deg"
    assert table.schema.field("wheel_center_z").metadata[b"unit"] == b"mm"
    assert table.column("wheel_center_z").to_pylist() == [3.0, 4.0]
    assert table.column("damper_length").to_pylist() == [None, None]


def test_writer_errors_match_the_reference():
    w = CsvWriter("/tmp/never.csv")
    with pytest.raises(ValueError, match="No frames to write"):
        w.write()
    w.add_frame(0, SolutionFrame({"a": (0.0, 0.0, 0.0)}, SolverInfo(True, 1, 0.0)))
    w.add_frame(1, SolutionFrame({"b": (0.0, 0.0, 0.0)}, SolverInfo(True, 1, 0.0)))
    with pytest.raises(ValueError, match="Frame 1 has inconsistent columns"):
        w.write()
    w2 = CsvWriter("/tmp/never.csv")
    w2.add_frame(0, SolutionFrame({"a": (0.0, 0.0, 0.0)}, SolverInfo(True, 1, 0.0), {"camber": 1.0}, {"camber": "deg"}))
    with pytest.raises(ValueError, match="Conflicting units"):
        w2.add_frame(1, SolutionFrame({"a": (0.0, 0.0, 0.0)}, SolverInfo(True, 1, 0.0), {"camber": 1.0}, {"camber": "mm"}))


def test_flatten_positions_uses_public_names():
    from open_kinematics_amd.enums import PointID
    from open_kinematics_amd.state import Point3

    flat = flatten_positions({PointID.WHEEL_CENTER: Point3([1, 2, 3])}, [PointID.WHEEL_CENTER, PointID.AXLE_INBOARD])
    assert flat == {"wheel_center": (1.0, 2.0, 3.0)}


def _spec_frame():
    """The frame of the reference's tests/test_metric_export_metadata.py:17-27, built from the drop-in's own types."""
    from open_kinematics_amd.results_writer import MetricKind, MetricSpec, MetricUnit, Scope

    spec = MetricSpec("camber", "Camber", MetricUnit.DEG, MetricKind.STATE, Scope.CORNER)
    return SolutionFrame(positions={"wheel_center": (1.0, 2.0, 3.0)}, solver_info=SolverInfo(True, 3, 1e-8),
                         metrics={"camber": -1.25}, metric_specs={"camber": spec})


def test_csv_writes_explicit_column_unit_metadata(tmp_path):
    """tests/test_metric_export_metadata.py:30-45 of the reference, against the drop-in writer."""
    output = tmp_path / "result.csv"
    writer = CsvWriter(output)
    writer.add_frame(0, _spec_frame())
    writer.write()
    units_line = next(line for line in output.read_text().splitlines() if line.startswith("# column_units:"))
    units = json.loads(units_line.partition(":")[2].strip())
    assert units["camber"] == "deg"
    assert units["wheel_center_z"] == "mm"


def test_parquet_writes_units_on_arrow_fields(tmp_path):
    """tests/test_metric_export_metadata.py:48-58 of the reference, against the drop-in writer."""
    import pyarrow.parquet as pq

    output = tmp_path / "result.parquet"
    writer = ParquetWriter(output)
    writer.add_frame(0, _spec_frame())
    writer.write()
    schema = pq.read_schema(output)
    assert schema.field("camber").metadata == {b"unit": b"deg"}
    assert schema.field("wheel_center_x").metadata == {b"unit": b"mm"}


def test_bulk_table_equals_the_per_frame_path(golden, tmp_path):
    """ResultTable.from_batch (arrays in, no per-row objects) writes the same bytes as frames through add_frame."""
    from open_kinematics_amd.results_writer import ResultTable, provenance

    arrays, program = golden("e2e_sweep")
    pos = arrays["ref_default_pos"]
    info = np.zeros(pos.shape[0], dtype=[("flags", "<i4"), ("nfev", "<i4"), ("max_residual", "<f8")])
    info["flags"], info["nfev"], info["max_residual"] = 1, arrays["ref_default_nfev"], arrays["ref_default_maxres"]
    info["flags"][3] = 3  # a step whose residual exceeded the tolerance: not converged
    metrics = {"camber": np.linspace(-2.0, 1.0, pos.shape[0]), "damper_length": np.where(np.arange(pos.shape[0]) % 2, np.nan, 300.5)}
    meta = provenance(tool="t")
    table = ResultTable.from_batch(program, pos, info, metrics)
    table.write_csv(tmp_path / "bulk.csv", meta)
    writer = CsvWriter(tmp_path / "frames.csv")
    writer.metadata = meta
    for b, frame in enumerate(frames_from_batch(program, pos, info, metrics)):
        writer.add_frame(b, frame)
    writer.write()
    assert (tmp_path / "bulk.csv").read_bytes() == (tmp_path / "frames.csv").read_bytes()
    rows = list(csv.DictReader(ln for ln in (tmp_path / "bulk.csv").read_text().splitlines() if not ln.startswith("#")))
    assert rows[3]["solver_converged"] == "False" and rows[2]["solver_converged"] == "True"
    assert rows[1]["damper_length"] == "" and rows[0]["damper_length"] == "300.5"
    table.write_parquet(tmp_path / "bulk.parquet", meta)
    import pyarrow.parquet as pq

    back = pq.read_table(tmp_path / "bulk.parquet")
    assert back.column_names == list(table.columns)
    assert back.column("damper_length").null_count == pos.shape[0] // 2
    assert back.schema.field("solver_nfev").type == "int64" and back.schema.field("solver_converged").type == "bool"


def test_a_million_row_ensemble_goes_to_parquet_in_seconds(tmp_path):
    """BASELINE config 5's size (4096 geometries x 256 steps, 15 output points): arrays -> Parquet without per-row objects."""
    import time

    from open_kinematics_amd.results_writer import write_batch
    from open_kinematics_amd.workloads import bump_sweep_problem

    program, _ = bump_sweep_problem(2)
    n = 4096 * 256
    rng = np.random.default_rng(0)
    pos = rng.normal(size=(n, program.n_out, 3))
    info = np.zeros(n, dtype=[("flags", "<i4"), ("nfev", "<i4"), ("max_residual", "<f8")])
    info["flags"], info["nfev"] = 1, 3
    metrics = {"camber": rng.normal(size=n), "caster": rng.normal(size=n)}
    t0 = time.perf_counter()
    write_batch(tmp_path / "c5.parquet", program, pos, info, metrics)
    elapsed = time.perf_counter() - t0
    import pyarrow.parquet as pq

    meta = pq.read_metadata(tmp_path / "c5.parquet")
    assert meta.num_rows == n and meta.num_columns == 4 + 2 + 3 * program.n_out
    assert elapsed < 5.0, f"{elapsed:.1f} s"


def test_none_is_missing_and_nan_is_a_value_like_the_reference(tmp_path):
    """cli/io/results_writer.py:316-335: None -> empty cell / null; a NaN stays a NaN; a column of nothing but None is a
    bool column of nulls; strings mixed with numbers fall back to a string column."""
    import pyarrow.parquet as pq

    frames = [
        SolutionFrame({"p": (0.0, 0.0, 0.0)}, SolverInfo(False, 7, float("nan")), {"camber": None, "never": None, "mixed": "left", "anti": 1.5}),
        SolutionFrame({"p": (0.0, 0.0, 1.0)}, SolverInfo(True, 3, 1e-9), {"camber": -0.5, "never": None, "mixed": 2, "anti": float("nan")}),
    ]
    for cls, path in ((CsvWriter, tmp_path / "out.csv"), (ParquetWriter, tmp_path / "out.parquet")):
        writer = cls(path, None, None)
        for k, frame in enumerate(frames):
            writer.add_frame(k, frame)
        writer.write()
    rows = [r for r in csv.reader(l for l in open(tmp_path / "out.csv") if not l.startswith("#"))]
    head, first, second = rows[0], dict(zip(rows[0], rows[1])), dict(zip(rows[0], rows[2]))
    assert first["solver_max_residual"] == "nan" and second["solver_max_residual"] == "1e-09"   # a failed solve's NaN is a value
    assert first["camber"] == "" and second["camber"] == "-0.5"                                  # None is a missing cell
    assert first["never"] == "" and second["never"] == ""
    assert first["mixed"] == "left" and second["mixed"] == "2"
    assert first["anti"] == "1.5" and second["anti"] == "nan"
    table = pq.read_table(tmp_path / "out.parquet")
    assert str(table.schema.field("never").type) == "bool" and table["never"].null_count == 2
    assert str(table.schema.field("mixed").type) == "string" and table["mixed"].to_pylist() == ["left", "2"]
    assert table["camber"].to_pylist() == [None, -0.5]
    residual = table["solver_max_residual"].to_pylist()
    assert residual[0] != residual[0] and table["solver_max_residual"].null_count == 0
    anti = table["anti"].to_pylist()
    assert anti[0] == 1.5 and anti[1] != anti[1] and table["anti"].null_count == 0
    assert head[:4] == ["step_index", "solver_converged", "solver_max_residual", "solver_nfev"]
