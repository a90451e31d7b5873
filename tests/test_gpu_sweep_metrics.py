"""GPU: compute_sweep_metrics drop-in (core/sweep.py:144-173) — row keys, order and values against the reference."""

import numpy as np
import pytest
import yaml

from conftest import gpu_available
from test_metrics_oracle import load_metrics_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _states(suspension, golden_pos, out_points):
    """SuspensionState objects holding the golden's output positions (what solve_sweep returned in the reference)."""
    from open_kinematics_amd.state import Point3, SuspensionState

    base = suspension.initial_state()
    states = []
    for row in golden_pos:
        positions = {k: v.copy() for k, v in base.positions.items()}
        for key, xyz in zip(out_points, row):
            positions[key] = Point3(xyz)
        states.append(SuspensionState(positions, set(base.free_points)))
    return states


def _close(got, ref, tol):
    if got is None or ref != ref:
        return got is None and ref != ref
    return abs(got - ref) <= tol * max(1.0, abs(ref))


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid"])
def test_corner_rows_match_the_reference(golden, name):
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import CATALOG_ORDER, METRIC_NAMES
    from open_kinematics_amd.sweep import compute_sweep_metrics

    arrays, _ = golden(name)
    mg = load_metrics_golden(name)
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    pick = range(0, mg["pos"].shape[0], max(1, mg["pos"].shape[0] // 20))
    states = _states(sus, mg["pos"][list(pick)], sus.output_points())
    result = compute_sweep_metrics(sus, sweep, states)  # the metrics depend on the states and target directions only
    assert result.derivative_error is None and len(result.rows) == len(states)
    keys = list(result.rows[0])
    assert keys[: len(CATALOG_ORDER)] == list(CATALOG_ORDER)                          # catalog.py:71-146
    assert [k for k in keys if k.startswith("deriv_")] == [str(n) for n in mg["deriv_names"]]  # same columns, same order
    for row, s in zip(result.rows, pick):
        for n in CATALOG_ORDER:
            assert _close(row[n], mg["values"][s][METRIC_NAMES.index(n)], 1e-9), (s, n)
        for j, col in enumerate(str(n) for n in mg["deriv_names"]):
            assert _close(row[col], mg["deriv"][s][j], 1e-7), (s, col)


@pytest.mark.parametrize("name,metrics_name,stride", [
    ("c3_axle_grid", "axle_c3", 4),                    # U-bar: arm angles + twist
    ("t_axle_t_bar_roll", "axle_t_bar_roll", 1),       # rigid T-bar, wheels opposed: twist
    ("t_axle_t_bar_bump", "axle_t_bar_bump", 1),       # rigid T-bar, wheels in phase: stem heave angle
    ("t_axle_heave_link", "axle_heave_link", 1),       # U-bar + rocker-to-rocker heave link (66 variables)
])
def test_axle_rows_match_the_reference(golden, name, metrics_name, stride):
    from open_kinematics_amd.enums import Side
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.metrics import AXLE_METRIC_NAMES, CATALOG_ORDER, METRIC_NAMES
    from open_kinematics_amd.sweep import compute_sweep_metrics

    arrays, _ = golden(name)
    mg = load_metrics_golden(metrics_name)
    axle = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), axle)
    pick = list(range(0, mg["pos"].shape[0], stride))
    states = _states(axle, mg["pos"][pick], axle.output_points())
    result = compute_sweep_metrics(axle, sweep, states)
    assert result.derivative_error is None
    first = result.rows[0]
    assert list(first.axle) == list(AXLE_METRIC_NAMES) + [str(n) for n in mg["axle_extra_names"]] + [str(n) for n in mg["axle_deriv_names"]]
    for tag, side in (("left", Side.LEFT), ("right", Side.RIGHT)):
        extras = [str(n) for n in mg[f"{tag}_extra_names"]]
        derivs = [str(n) for n in mg[f"{tag}_deriv_names"]]
        # reference order: catalog, actuation / spring values, derivative columns, then a U-bar's arm angle
        tail = [n for n in extras if n == "arb_arm_angle"]
        assert list(first.corners[side]) == list(CATALOG_ORDER) + [n for n in extras if n not in tail] + derivs + tail
    if "axle_key_order" in mg:
        assert list(first.axle) == [str(n) for n in mg["axle_key_order"]]
        assert list(first.corners[Side.LEFT]) == [str(n) for n in mg["left_key_order"]]
    for row, s in zip(result.rows, pick):
        for k, n in enumerate(AXLE_METRIC_NAMES):
            assert _close(row.axle[n], mg["axle_values"][s][k], 1e-9), (s, n)
        for j, n in enumerate(str(x) for x in mg["axle_extra_names"]):  # arb_twist, t_bar_heave_angle, heave_link_length
            assert _close(row.axle[n], mg["axle_extra_values"][s][j], 1e-9), (s, n)
        for j, col in enumerate(str(n) for n in mg["axle_deriv_names"]):
            assert _close(row.axle[col], mg["axle_deriv"][s][j], 1e-7), (s, col)
        for tag, side in (("left", Side.LEFT), ("right", Side.RIGHT)):
            corner = row.corners[side]
            for n in CATALOG_ORDER:
                assert _close(corner[n], mg[f"{tag}_values"][s][METRIC_NAMES.index(n)], 1e-9), (s, tag, n)
            for j, n in enumerate(str(x) for x in mg[f"{tag}_extra_names"]):
                assert _close(corner[n], mg[f"{tag}_extra_values"][s][j], 1e-9), (s, tag, n)
            for j, n in enumerate(str(x) for x in mg[f"{tag}_deriv_names"]):
                assert _close(corner[n], mg[f"{tag}_deriv"][s][j], 1e-7), (s, tag, n)
    flat = first.flat_row()
    assert "camber_left" in flat and "arb_twist" in flat and list(flat)[-1] == str(mg["axle_deriv_names"][-1])


def test_end_to_end_csv_reproduces_the_reference_file(tmp_path):
    """geometry.yaml + sweep.yaml -> solve_sweep -> compute_sweep_metrics -> CsvWriter: the reference's committed
    e2e output (tests/data/e2e/output.csv) column for column — header, order, units, values (other platform: 5e-5)."""
    import csv
    import json
    import os

    from conftest import GOLDEN
    from open_kinematics_amd.input import load_geometry, load_sweep
    from open_kinematics_amd.results_writer import CsvWriter, frames_from_states
    from open_kinematics_amd.sweep import compute_sweep_metrics, solve_sweep

    geometry_path = os.path.join(GOLDEN, "geometry", "geometry.yaml")
    sweep_path = os.path.join(GOLDEN, "geometry", "sweep.yaml")
    sus = load_geometry(geometry_path)
    sweep = load_sweep(sweep_path, sus)
    states, infos = solve_sweep(sus, sweep)
    metrics = compute_sweep_metrics(sus, sweep, states)
    assert metrics.derivative_error is None
    out = tmp_path / "out.csv"
    writer = CsvWriter(out, geometry_path=geometry_path, sweep_path=sweep_path)
    for k, frame in enumerate(frames_from_states(states, infos, metrics.rows, sus.output_points())):
        writer.add_frame(k, frame)
    writer.write()

    def read(path):
        lines = open(path, encoding="utf-8").read().splitlines()
        meta = [ln for ln in lines if ln.startswith("#")]
        return meta, list(csv.DictReader(ln for ln in lines if not ln.startswith("#")))

    ref_meta, ref_rows = read(os.path.join(GOLDEN, "e2e_output.csv"))
    my_meta, my_rows = read(out)
    assert list(my_rows[0].keys()) == list(ref_rows[0].keys())            # all 76 columns, same order
    units = lambda meta: json.loads(next(ln for ln in meta if ln.startswith("# column_units")).split(": ", 1)[1])  # noqa: E731
    assert units(my_meta) == units(ref_meta)
    assert len(my_rows) == len(ref_rows)
    far = ("svic_x", "svic_z", "svsa_length", "fvic_y", "fvic_z", "fvsa_length", "svsa_angle")  # ill-conditioned: relative
    for mine, ref in zip(my_rows, ref_rows):
        for col, ref_cell in ref.items():
            cell = mine[col]
            if col in ("step_index", "solver_converged"):
                assert cell == ref_cell
            elif col in ("solver_nfev", "solver_max_residual"):
                continue  # solver-path dependent (the reference's own e2e test excludes them, tests/e2e/test_e2e.py:37-39)
            elif ref_cell == "":
                assert cell == "", col
            else:
                a, b = float(cell), float(ref_cell)
                tol = 2e-3 * max(1.0, abs(b)) if col in far else 5e-5 * max(1.0, abs(b))
                assert abs(a - b) <= tol, (col, a, b)
