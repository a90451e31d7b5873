#!/usr/bin/env python3
"""
What an EVALUATED state costs: solve -> tangents -> metric catalog with derivative columns as three launches against
the one launch of okx_solve_evaluated_batch (the epilogue of the solve kernels), HIP events, on

  c5   BASELINE config 5: 4096 perturbed double-wishbone geometries x 256 steps (1 048 576 states, lane kernels,
       per-geometry tables)
  c2   BASELINE config 2: one 16384-step sweep of the program's own geometry (quad cold body)
  c4   the MacPherson 512 x 512 grid (lane kernels, own geometry)

  python tools/evaluated_rates.py [c5 c2 c4] [--reps 20] [--only evaluated_metrics_only]   (one row: what
                                                                 tools/profile_run.sh wraps in rocprofv3)
"""

from __future__ import annotations

import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def ev_ms(fn, device, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / reps


def measure(which: str, device, reps: int, only: str = "") -> dict:
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles, corner_state_metrics
    from open_kinematics_amd.workloads import bump_sweep_problem, ensemble_problem, geometry_path, macpherson_grid_problem

    kw = {}
    if which == "c5":
        program, table, rel = ensemble_problem()
        dp = DeviceProgram(program, device)
        gpos, grow = dp.rebind(torch.as_tensor(table, device=device))
        targets = dp.ensemble_targets(gpos, rel)
        kw = dict(geom_pos=gpos, geom_row_param=grow, steps_per_geometry=rel.shape[0])
        sus = load_geometry(geometry_path("geometry.yaml"))
    elif which == "c4":
        program, t = macpherson_grid_problem()
        dp = DeviceProgram(program, device)
        targets = torch.as_tensor(t, device=device)
        sus = load_geometry(geometry_path("macpherson_geometry.yaml"))
    else:
        program, t = bump_sweep_problem(16384)
        dp = DeviceProgram(program, device)
        targets = torch.as_tensor(t, device=device)
        sus = load_geometry(geometry_path("geometry.yaml"))
    roles = corner_roles(sus, program)
    dp.enable_evaluation(roles)
    n = targets.shape[0]
    T = program.n_targets
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    evb = torch.empty((n, 1 + T, 24), dtype=torch.float64, device=device)
    skw = dict(chain_len=1, predictor=False, **kw)
    rows = {}
    if only == "evaluated_metrics_only":
        rows[only] = ev_ms(dp.plan_evaluated(targets, info_out=info, eval_out=evb, output="none", **skw), device, reps)
        return {"workload": which, "states": n, "ms": rows, "states_per_s": {k: n / v * 1e3 for k, v in rows.items()},
                "roofline": {"kernel_ms": rows[only]},  # (the field tools/save_profile.py reads)
                "algorithmic_bytes_per_state": 8 * T + 8 * 24 * (1 + T) + 16}
    solve = dp.plan(targets, out=out, info_out=info, **skw)
    if only == "evaluate_given_states":  # okx_evaluate_batch alone (its lane form on a batch that fills the chip)
        solve()
        rows[only] = ev_ms(lambda: dp.evaluate(out, eval_out=evb, **kw), device, reps)
        return {"workload": which, "states": n, "ms": rows, "states_per_s": {k: n / v * 1e3 for k, v in rows.items()},
                "roofline": {"kernel_ms": rows[only]},
                # what the kernel must move: the FREE points of a record (fixed points come with the geometry, derived points
                # are re-evaluated) in, the evaluation rows out; `with_whole_records`: had it to read all 24 n_out bytes
                "algorithmic_bytes_per_state": 24 * program.n_free + 8 * 24 * (1 + T),
                "with_whole_records_bytes_per_state": 24 * program.n_out + 8 * 24 * (1 + T)}
    rows["solve_records"] = ev_ms(solve, device, reps)
    rows["solve_output_none"] = ev_ms(dp.plan(targets, info_out=info, output="none", **skw), device, reps)
    solve()
    tan, _ = dp.tangents(out, **kw)
    rows["tangents"] = ev_ms(lambda: dp.tangents(out, **kw), device, reps)
    rows["metrics_with_derivatives"] = ev_ms(lambda: corner_state_metrics(roles, out, tan), device, reps)
    rows["three_launches"] = rows["solve_records"] + rows["tangents"] + rows["metrics_with_derivatives"]
    rows["evaluated_metrics_only"] = ev_ms(dp.plan_evaluated(targets, info_out=info, eval_out=evb, output="none", **skw), device, reps)
    rows["evaluated_with_records"] = ev_ms(dp.plan_evaluated(targets, out=out, info_out=info, eval_out=evb, **skw), device, reps)
    del tan
    rows["evaluate_given_states"] = ev_ms(lambda: dp.evaluate(out, eval_out=evb, **kw), device, reps)
    os.environ["OKX_DEV"] = "evaluate_quad"   # (the quad form, 16 states per wavefront: what small batches get)
    rows["evaluate_given_states_quad_form"] = ev_ms(lambda: dp.evaluate(out, eval_out=evb, **kw), device, reps)
    os.environ["OKX_DEV"] = "evaluate_lane"
    rows["evaluate_given_states_lane_form"] = ev_ms(lambda: dp.evaluate(out, eval_out=evb, **kw), device, reps)
    del os.environ["OKX_DEV"]
    res = {"workload": which, "states": n, "evaluation": dp.evaluation, "evaluation_note": dp.evaluation_note, "ms": rows,
           "states_per_s": {k: n / v * 1e3 for k, v in rows.items()},
           "bytes_per_state_out": {"evaluated_metrics_only": 8 * 24 * (1 + T) + 40 + 8 * T,
                                   "three_launches": 24 * program.n_out * (2 + 2 * T) + (1 + T) * 152 + 64}}
    return res


def main():
    args = sys.argv[1:]
    reps, only = 20, ""
    for flag in ("--reps", "--only"):
        if flag in args:
            at = args.index(flag)
            value = args[at + 1]
            del args[at:at + 2]
            if flag == "--reps":
                reps = int(value)
            else:
                only = value
    which = args or ["c5", "c2", "c4"]
    device = torch.device("cuda:0")
    for w in which:
        print(json.dumps(measure(w, device, reps, only)), flush=True)


if __name__ == "__main__":
    main()
