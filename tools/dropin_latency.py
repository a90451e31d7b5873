#!/usr/bin/env python3
"""
Wall time of the drop-in's sweep calls on the reference's own fixtures (one process, warm): solve_sweep,
compute_sweep_metrics on its states, and solve_evaluated_sweep (corners: ONE launch for solve + tangents + metrics;
axles: solve, then evaluate) - what a user of kinematics.core.sweep sees per call.

  python tools/dropin_latency.py [fixture ...]      (tests/golden/<fixture>.npz holding geometry_yaml / sweep_yaml)
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import yaml  # noqa: E402


def main():
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.sweep import compute_sweep_metrics, solve_evaluated_sweep, solve_sweep

    forced = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--segment=")]
    if forced:  # another chain length for the parallel chains of a warm-started sweep (solver._segment_length)
        from open_kinematics_amd import solver as _solver
        _solver._segment_length = lambda n_steps: forced[0] if n_steps >= 4 else 0
    names = [a for a in sys.argv[1:] if not a.startswith("--segment=")] or ["c1_dw_corner", "c4_macpherson_grid", "t_corner_rocker", "e2e_sweep", "t_axle_dw", "t_axle_macpherson", "t_axle_heave_link", "t_axle_t_bar_bump", "t_axle_t_bar_roll", "t_axle_t_bar_heave"]
    for name in names:
        path = os.path.join(REPO, "tests", "golden", name + ".npz")
        if not os.path.exists(path):
            print(json.dumps({"fixture": name, "error": "no such golden"}))
            continue
        arrays = dict(np.load(path, allow_pickle=False))
        if "geometry_yaml" not in arrays or "sweep_yaml" not in arrays:
            print(json.dumps({"fixture": name, "error": "no yaml in this golden"}))
            continue
        sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
        sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)

        def best(fn, reps=5):
            fn()
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                out = fn()
                times.append(time.perf_counter() - t0)
            return min(times) * 1e3, out

        solve_ms, (states, stats) = best(lambda: solve_sweep(sus, sweep))
        metrics_ms, _ = best(lambda: compute_sweep_metrics(sus, sweep, states))
        evaluated_ms, ev = best(lambda: solve_evaluated_sweep(sus, sweep))
        fused_ms, _ = best(lambda: solve_evaluated_sweep(sus, sweep, fused=True))
        print(json.dumps({"fixture": name, "steps": len(states), "axle": hasattr(sus, "corners"), "metric_columns": len(ev.metrics.rows[0]) if isinstance(ev.metrics.rows[0], dict) else None,
                          "solve_sweep_ms": round(solve_ms, 3), "compute_sweep_metrics_ms": round(metrics_ms, 3),
                          "solve_evaluated_sweep_ms": round(evaluated_ms, 3),
                          "solve_evaluated_sweep_one_launch_ms": round(fused_ms, 3)}), flush=True)


if __name__ == "__main__":
    main()
