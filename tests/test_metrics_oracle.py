"""CPU: numpy restatement of the corner state metrics / derivative columns against the reference's outputs."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.metrics_oracle import (
    AXLE_METRIC_NAMES,
    METRIC_NAMES,
    axle_metrics,
    corner_metrics,
    geometry_metrics,
    rotation_about_fixed_axis_deg,
)
from oracle.oracle import Oracle

FIXTURES = ["c1_dw_corner", "c4_macpherson_grid", "e2e_sweep"]
# golden name -> base fixture whose program / output points the states belong to
ANTI_FIXTURES = {"dw_front_anti": "c1_dw_corner", "mac_rear_anti": "c4_macpherson_grid"}


def close(got, ref, rel=1e-9):
    """NaN where the reference reports None, else |got - ref| <= rel * max(1, |ref|)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(ref)):
        return False
    ok = ~np.isnan(ref)
    return bool(np.all(np.abs(got[ok] - ref[ok]) <= rel * np.maximum(1.0, np.abs(ref[ok]))))


def geometry_kwargs(names, mg, prefix=""):
    """Instant-axis / damper indices and vehicle numbers of a metrics golden (names = output point names)."""
    text = lambda key: str(mg[prefix + key])  # noqa: E731
    damper = [names.index(str(n)) for n in mg[prefix + "damper"]]
    bias = float(mg[prefix + "front_brake_bias"])
    return dict(
        axis_kind=text("axis_kind"), axis_idx=[names.index(str(n)) for n in mg[prefix + "axis_points"]],
        damper_idx=damper or None, wheelbase=float(mg[prefix + "wheelbase"]), cg_z=float(mg[prefix + "cg_z"]),
        front_brake_bias=None if np.isnan(bias) else bias, axle_position=text("axle_position") or None,
        driven_axle=text("driven_axle") or None,
    )


def oracle_geometry_row(pos_row, roles, side, g):
    return geometry_metrics(
        {"wheel_center": pos_row[roles["wheel_center"]], "contact_patch": pos_row[roles["contact_patch"]]}, side,
        g["axis_kind"], [pos_row[i] for i in g["axis_idx"]],
        damper=None if g["damper_idx"] is None else (pos_row[g["damper_idx"][0]], pos_row[g["damper_idx"][1]]),
        wheelbase=g["wheelbase"], cg_z=g["cg_z"], front_brake_bias=g["front_brake_bias"],
        axle_position=g["axle_position"], driven_axle=g["driven_axle"])


def load_metrics_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"metrics_{name}.npz"), allow_pickle=False))


def role_indices(program, mg):
    """Output-list index of every role point; names from the reference's role hooks."""
    return role_indices_by_name(out_names(program), mg)


def out_names(program):
    return [program.point_keys[k].lower_name for k in program.out_point]


def role_indices_by_name(names, mg, prefix="", side=""):
    axle_in, axle_out, lower, upper = (side + str(v).lower() for v in mg[prefix + "roles"])
    return {"wheel_center": names.index(side + "wheel_center"), "contact_patch": names.index(side + "contact_patch_center"),
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
    geometry = geometry_kwargs(out_names(program), mg)
    if geometry["damper_idx"] is not None:
        roles = {**roles, "damper_top": geometry["damper_idx"][0], "damper_bottom": geometry["damper_idx"][1]}
    orc = Oracle(program)  # softnorm rows + pins: the reference's own tangent formulation
    free_out = [list(program.out_point).index(int(p)) for p in program.free_point]
    steps = range(0, mg["pos"].shape[0], max(1, mg["pos"].shape[0] // 16))
    for s in steps:
        pos = {k: mg["pos"][s][i] for k, i in roles.items()}
        values, _ = corner_metrics(pos, None, side, design_z)
        assert np.max(np.abs(values[:8] - mg["values"][s][:8])) <= 1e-10
        assert close(oracle_geometry_row(mg["pos"][s], roles, side, geometry), mg["values"][s][8:], 1e-11), s
        vel, _, _ = orc.tangents(mg["pos"][s][free_out].reshape(-1))
        vel = vel[:, program.out_point]
        for j, (what, t) in plan.items():
            if isinstance(what, tuple):
                got = vel[t][roles["wheel_center"]][what[1]]
            else:
                _, d = corner_metrics(pos, {k: vel[t][i] for k, i in roles.items()}, side, design_z)
                got = d[what]
            assert abs(got - mg["deriv"][s][j]) <= 1e-9 * max(1.0, abs(mg["deriv"][s][j])), (s, mg["deriv_names"][j])


@pytest.mark.parametrize("name", sorted(ANTI_FIXTURES))
def test_anti_geometry_oracle_matches_the_reference(golden, name):
    """Goldens authored with brake bias / axle position / driven axle: anti-dive, anti-lift, anti-squat defined."""
    _, program = golden(ANTI_FIXTURES[name])
    mg = load_metrics_golden(name)
    names = out_names(program)
    roles = role_indices(program, mg)
    geometry = geometry_kwargs(names, mg)
    side = float(mg["side_sign"])
    defined = np.isfinite(mg["values"]).all(axis=0)
    expect = {"dw_front_anti": ("anti_dive", "anti_squat"), "mac_rear_anti": ("anti_lift", "anti_squat", "damper_length")}
    assert all(defined[METRIC_NAMES.index(n)] for n in expect[name])
    for s in range(mg["pos"].shape[0]):
        assert close(oracle_geometry_row(mg["pos"][s], roles, side, geometry), mg["values"][s][8:], 1e-11), s


def axle_sides(names, mg, pos_row, design):
    """Inputs of ``axle_metrics`` for one state of the axle golden."""
    sides = {}
    for tag in ("left", "right"):
        wc, cp = names.index(f"{tag}_wheel_center"), names.index(f"{tag}_contact_patch_center")
        rack = str(mg[f"{tag}_rack"])
        sides[tag] = dict(
            wheel_center=pos_row[wc], contact_patch=pos_row[cp], design_wheel_center_z=float(design[wc][2]),
            design_contact_patch_z=float(design[cp][2]), axis_kind=str(mg[f"{tag}_axis_kind"]),
            axis_points=[pos_row[names.index(f"{tag}_{n}")] for n in mg[f"{tag}_axis_points"]],
            rack_y=None if not rack else float(pos_row[names.index(f"{tag}_{rack}")][1]),
            design_rack_y=0.0 if not rack else float(design[names.index(f"{tag}_{rack}")][1]),
        )
    return sides


def test_axle_metric_oracle_matches_the_reference(golden):
    _, program = golden("c3_axle_grid")
    mg = load_metrics_golden("axle_c3")
    names = out_names(program)
    design = program.design_pos[program.out_point]
    assert len(AXLE_METRIC_NAMES) == mg["axle_values"].shape[1]
    for s in range(mg["pos"].shape[0]):
        assert close(axle_metrics(axle_sides(names, mg, mg["pos"][s], design)), mg["axle_values"][s], 1e-10), s
        for tag in ("left", "right"):
            roles = role_indices_by_name(names, mg, f"{tag}_", f"{tag}_")
            side = float(mg[f"{tag}_side_sign"])
            geometry = geometry_kwargs([n[len(tag) + 1:] if n.startswith(tag + "_") else "-" for n in names], mg, f"{tag}_")
            pos = {k: mg["pos"][s][i] for k, i in roles.items()}
            values, _ = corner_metrics(pos, None, side, float(design[roles["wheel_center"]][2]))
            assert np.max(np.abs(values[:8] - mg[f"{tag}_values"][s][:8])) <= 1e-10
            assert close(oracle_geometry_row(mg["pos"][s], roles, side, geometry), mg[f"{tag}_values"][s][8:], 1e-11)


def axle_rotation_specs(program):
    """name -> (output index of the moving pickup, design position, axis point, unit axis, scale) of the C3 axle."""
    pk = [program.point_keys[k].lower_name for k in range(program.n_points)]
    names = out_names(program)
    design = program.design_pos
    specs = {}

    def axis(a, b):
        pa, pb = design[pk.index(a)], design[pk.index(b)]
        return pa, (pb - pa) / np.linalg.norm(pb - pa)

    for tag, sign in (("left", 1.0), ("right", -1.0)):
        a, u = axis(f"{tag}_rocker_axis_a", f"{tag}_rocker_axis_b")
        specs[f"rocker_angle_{tag}"] = (names.index(f"{tag}_pushrod_inboard"), design[pk.index(f"{tag}_pushrod_inboard")], a, u, sign)
        specs[f"torsion_bar_twist_{tag}"] = specs[f"rocker_angle_{tag}"]
        a, u = axis("center_arb_u_bar_axis_a", "center_arb_u_bar_axis_b")
        specs[f"arb_arm_angle_{tag}"] = (names.index(f"{tag}_droplink_u_bar"), design[pk.index(f"{tag}_droplink_u_bar")], a, u, 1.0)
    return specs


def test_topology_rotation_metrics_match_the_reference(golden):
    """rocker_angle / torsion_bar_twist / arb_arm_angle per corner, arb_twist per axle (values)."""
    _, program = golden("c3_axle_grid")
    mg = load_metrics_golden("axle_c3")
    specs = axle_rotation_specs(program)
    for s in range(mg["pos"].shape[0]):
        angle = {k: rotation_about_fixed_axis_deg(mg["pos"][s][o], None, d, a, u, sc)[0] for k, (o, d, a, u, sc) in specs.items()}
        for tag in ("left", "right"):
            for j, name in enumerate(str(n) for n in mg[f"{tag}_extra_names"]):
                assert abs(angle[f"{name}_{tag}"] - mg[f"{tag}_extra_values"][s][j]) <= 1e-10, (s, tag, name)
        assert [str(n) for n in mg["axle_extra_names"]] == ["arb_twist"]
        assert abs(angle["arb_arm_angle_left"] - angle["arb_arm_angle_right"] - mg["axle_extra_values"][s][0]) <= 1e-10


def test_e2e_csv_metric_columns_are_the_same_numbers():
    """The reference's committed e2e CSV carries the same catalog columns (written on another platform)."""
    import csv

    mg = load_metrics_golden("e2e_sweep")
    with open(os.path.join(GOLDEN, "e2e_output.csv"), encoding="utf-8") as fh:
        rows = list(csv.DictReader(line for line in fh if not line.startswith("#")))
    assert len(rows) == mg["values"].shape[0]
    for k, n in enumerate(METRIC_NAMES):
        col = np.array([float(r[n]) if r[n] != "" else np.nan for r in rows])
        if k < 8:
            assert np.max(np.abs(col - mg["values"][:, k])) <= 5e-5, n  # states agree to 1.4e-5 mm across platforms
        else:  # instant centres far from the car amplify that state difference: compare relatively
            assert close(col, mg["values"][:, k], 2e-3), n
