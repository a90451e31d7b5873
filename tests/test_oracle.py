"""
Pins the CPU oracle (oracle/okx_oracle.c) against outputs of the REAL reference that
oracle/gen_golden.py captured in this container (tests/golden/*.npz), and against the
reference's own e2e golden CSV.  CPU only.
"""

import csv
import os

import numpy as np
import pytest

from conftest import ALL_ROW_CLASSES, GOLDEN, STEERED, UNSTEERED
from oracle.oracle import Oracle


@pytest.mark.parametrize("name", STEERED + UNSTEERED + ALL_ROW_CLASSES)
def test_residual_and_jacobian_match_reference(golden, name):
    """R1: ResidualComputer.compute / compute_jacobian (solver.py:226-275, :502-581)."""
    arrays, program = golden(name)
    r, jac = Oracle(program).eval(arrays["eval_x"], arrays["eval_targets"])
    # a distance row is sqrt(|d|^2 + eps^2) - eps - L with |d| up to ~1 m: one ulp of the
    # length (2.3e-13 at 1024 mm) is the floor for any re-ordering of the 3-term sum
    assert np.all(np.abs(r - arrays["eval_r"]) <= 2.5e-13 + 1e-13 * np.abs(arrays["eval_r"]))
    assert np.all(np.abs(jac - arrays["eval_jac"]) <= 1e-13 * np.maximum(1.0, np.abs(arrays["eval_jac"])))
    # structure: columns the reference never touches stay exactly zero
    untouched = np.all(arrays["eval_jac"] == 0.0, axis=0)
    assert np.all(jac[:, untouched] == 0.0)


@pytest.mark.parametrize("name", STEERED)
def test_sweep_matches_reference_on_rack_steered(golden, name):
    """
    R3: same algorithm (MINPACK lmder, sequential warm start) on the rack-steered BASELINE
    configs.  The reference does not reproduce itself better than ~1.4e-5 mm (default
    tolerances) / ~3.5e-8 mm (tight) here because of the zero-gradient point-on-line row
    (SURVEY.md §8c ladder); the oracle has to sit inside that same band.
    """
    arrays, program = golden(name)
    orc = Oracle(program)
    res = orc.sweep(arrays["targets_abs"])
    assert res.first_failed_step == -1
    assert np.max(np.abs(res.positions - arrays["ref_default_pos"])) <= 5e-5
    assert abs(res.info["nfev"].mean() / arrays["ref_default_nfev"].mean() - 1.0) <= 0.10
    assert res.info["max_residual"].max() <= 1e-4
    tight = orc.sweep(arrays["targets_abs"], 1e-15, 1e-15, 1e-15)
    assert np.max(np.abs(tight.positions - arrays["ref_tight_pos"])) <= 1e-7
    # default vs tight of the reference itself, for the record of the ladder
    assert np.max(np.abs(arrays["ref_default_pos"] - arrays["ref_tight_pos"])) <= 5e-5


@pytest.mark.parametrize("name", UNSTEERED)
def test_sweep_matches_reference_without_degenerate_row(golden, name):
    """R2: no point-on-line row -> unique, well-conditioned minimiser: <= 1e-9 mm."""
    arrays, program = golden(name)
    orc = Oracle(program)
    res = orc.sweep(arrays["targets_abs"])
    assert res.first_failed_step == -1
    assert np.max(np.abs(res.positions - arrays["ref_default_pos"])) <= 1e-9
    assert np.array_equal(res.info["nfev"], arrays["ref_default_nfev"])
    tight = orc.sweep(arrays["targets_abs"], 1e-15, 1e-15, 1e-15)
    assert np.max(np.abs(tight.positions - arrays["ref_tight_pos"])) <= 1e-9
    cold = orc.sweep(arrays["targets_abs"], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert np.max(np.abs(cold.positions - arrays["ref_tight_pos"])) <= 1e-9


def test_pinned_line_rows_share_the_fixed_point(golden):
    """The line-pin form (okx.h OKX_ROW_LINE_PIN) has the same minimiser up to the
    reference's own convergence floor on the rack pickup (2.2e-8)."""
    for name in ("c1_dw_corner", "c4_macpherson_grid"):
        arrays, program = golden(name)
        pinned = program.with_line_mode("pinned")
        assert pinned.n_rows == program.n_rows + 2
        res = Oracle(pinned).sweep(arrays["targets_abs"][::8], 1e-15, 1e-15, 1e-15, warm_start=False)
        assert res.first_failed_step == -1
        diff = np.abs(res.positions - arrays["ref_tight_pos"][::8])
        assert diff.max() <= 6e-8
        rack = [i for i, k in enumerate(program.out_point)
                if program.point_keys[k].lower_name.endswith("trackrod_inboard")]
        others = np.delete(diff, rack, axis=1)
        assert others.max() <= 1e-8
        assert pinned.with_line_mode("softnorm").n_rows == program.n_rows


def test_e2e_golden_csv_of_the_reference(golden):
    """The reference's committed tests/data/e2e/output.csv (tests/e2e/test_e2e.py:316-364)."""
    arrays, program = golden("e2e_sweep")
    with open(os.path.join(GOLDEN, "e2e_output.csv"), "r", encoding="utf-8") as fh:
        lines = [ln for ln in fh if not ln.strip().startswith("#")]
    rows = list(csv.DictReader(lines))
    assert len(rows) == arrays["targets_abs"].shape[0] == 41
    res = Oracle(program).sweep(arrays["targets_abs"])
    worst = 0.0
    for k, idx in enumerate(program.out_point):
        name = program.point_keys[idx].lower_name
        for a, axis in enumerate("xyz"):
            col = np.array([float(r[f"{name}_{axis}"]) for r in rows])
            worst = max(worst, float(np.max(np.abs(col - res.positions[:, k, a]))))
    # the reference compares at atol=rtol=1e-3 (test_e2e.py:204-211); we are far inside
    assert worst <= 5e-5


def test_rebind_matches_reference_problem_emission(golden):
    """C5 / SURVEY H5: per-geometry design targets recomputed from perturbed hardpoints."""
    arrays, program = golden("c5_ensemble")
    orc = Oracle(program)
    for g in range(arrays["hardpoints"].shape[0]):
        pos, rp = orc.rebind(arrays["hardpoints"][g])
        assert np.max(np.abs(pos - arrays["design_pos"][g])) <= 1e-12
        ref = arrays["row_param"][g]
        assert np.max(np.abs(rp - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-14
    # solving the rebound problem reproduces the reference's solve of that geometry
    from open_kinematics_amd.program import ConstraintProgram

    g = 3
    pos, rp = orc.rebind(arrays["hardpoints"][g])
    rebound = ConstraintProgram.from_arrays(arrays, prefix="prog_")
    rebound.design_pos = pos
    rebound.row_param = rp
    res = Oracle(rebound).sweep(arrays["targets_abs"][g], 1e-15, 1e-15, 1e-15)
    assert np.max(np.abs(res.positions - arrays["ref_tight_pos"][g])) <= 1e-7


def test_underdetermined_is_rejected(golden):
    """solver.py:116-121 / tests/core/test_solver.py:198-207."""
    _, program = golden("u_dw_corner")
    stripped = program.with_targets([], np.zeros((0, 3)))
    stripped.row_type = stripped.row_type[:5]
    stripped.row_pts = stripped.row_pts[:5]
    stripped.row_param = stripped.row_param[:5]
    stripped.row_source = stripped.row_source[:5]
    with pytest.raises(ValueError, match="System is underdetermined"):
        Oracle(stripped)


def test_infeasible_target_is_flagged(golden):
    """solver.py:735-747: a lock-out target converges to a compromise and is rejected."""
    arrays, program = golden("u_dw_corner")
    targets = arrays["targets_abs"][:1].copy()
    targets[0, 0] += 2000.0  # wheel centre 2 m above design: unreachable
    res = Oracle(program).sweep(targets)
    assert res.first_failed_step == 0
    assert res.info["max_residual"][0] > 1e-3
