"""CPU: numpy restatement of the corner state metrics / derivative columns against the reference's outputs."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.metrics_oracle import METRIC_NAMES, corner_metrics
from oracle.oracle import Oracle

FIXTURES = ["c1_dw_corner", "c4_macpherson_grid", "e2e_sweep"]


def load_metrics_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"metrics_{name}.npz"), allow_pickle=False))


def role_indices(program, mg):
    """Output-list index of every role point; names from the reference's role hooks."""
    names = [program.point_keys[k].lower_name for k in program.out_point]
    axle_in, axle_out, lower, upper = (str(v).lower() for v in mg["roles"])
    return {"wheel_center": names.index("wheel_center"), "contact_patch": names.index("contact_patch_center"),
            "axle_inboard": names.index(axle_in), "axle_outboard": names.index(axle_out),
            "steer_lower": names.index(lower), "steer_upper": names.index(upper)}


def derivative_plan(program, deriv_names):
    """deriv column -> (metric index or ('wc', axis), target index)."""
    tnames = [program.point_keys[p].lower_name for p in program.tgt_point]
    hub = [t for t, (n, d) in enumerate(zip(tnames, program.tgt_dir)) if n == "wheel_center" and d[2] == 1.0]
    rack = [t for t, (n, d) in enumerate(zip(tnames, program.tgt_dir)) if n == "trackrod_inboard" and d[1] == 1.0]
    plan = {}
    for j, col in enumerate(str(c) for c in deriv_names):
        response, driver = col[len("deriv_"):].split("_wrt_")
        target = {"hub_z": hub, "rack_displacement": rack}.get(driver)
        if not target:
            continue
        if response in METRIC_NAMES:
            plan[j] = (METRIC_NAMES.index(response), target[0])
        elif response == "wheel_center_x":
            plan[j] = (("wc", 0), target[0])
    return plan


@pytest.mark.parametrize("name", FIXTURES)
def test_metric_oracle_matches_the_reference(golden, name):
    _, program = golden(name)
    mg = load_metrics_golden(name)
    roles = role_indices(program, mg)
    side = float(mg["side_sign"])
    design_z = float(program.design_pos[program.out_point[roles["wheel_center"]]][2])
    plan = derivative_plan(program, mg["deriv_names"])
    assert len(plan) >= 8
    orc = Oracle(program)  # softnorm rows + pins: the reference's own tangent formulation
    free_out = [list(program.out_point).index(int(p)) for p in program.free_point]
    steps = range(0, mg["pos"].shape[0], max(1, mg["pos"].shape[0] // 16))
    for s in steps:
        pos = {k: mg["pos"][s][i] for k, i in roles.items()}
        values, _ = corner_metrics(pos, None, side, design_z)
        assert np.max(np.abs(values - mg["values"][s])) <= 1e-10
        vel, _, _ = orc.tangents(mg["pos"][s][free_out].reshape(-1))
        vel = vel[:, program.out_point]
        for j, (what, t) in plan.items():
            if isinstance(what, tuple):
                got = vel[t][roles["wheel_center"]][what[1]]
            else:
                _, d = corner_metrics(pos, {k: vel[t][i] for k, i in roles.items()}, side, design_z)
                got = d[what]
            assert abs(got - mg["deriv"][s][j]) <= 1e-9 * max(1.0, abs(mg["deriv"][s][j])), (s, mg["deriv_names"][j])


def test_e2e_csv_metric_columns_are_the_same_numbers():
    """The reference's committed e2e CSV carries the same eight columns (other platform: 1e-6)."""
    import csv

    mg = load_metrics_golden("e2e_sweep")
    with open(os.path.join(GOLDEN, "e2e_output.csv"), encoding="utf-8") as fh:
        rows = list(csv.DictReader(line for line in fh if not line.startswith("#")))
    assert len(rows) == mg["values"].shape[0]
    for k, n in enumerate(METRIC_NAMES):
        col = np.array([float(r[n]) for r in rows])
        assert np.max(np.abs(col - mg["values"][:, k])) <= 5e-5, n  # states agree to 1.4e-5 mm across platforms
