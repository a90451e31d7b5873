#!/usr/bin/env python3
"""The drop-in's default (a warm-started sweep solved as cold starts side by side, kept when it is the sequential path) against
the sequential chain on PERTURBED copies of a fixture's geometry: worst difference of any point, and how many sweeps fell
back to the chain.
  python tools/dropin_policy_check.py [fixture] [n geometries] [sigma mm]"""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import yaml  # noqa: E402


def perturb(node, rng, sigma):
    """Every [x, y, z] triple of numbers in the geometry document moved by N(0, sigma)."""
    if isinstance(node, dict):
        return {k: perturb(v, rng, sigma) for k, v in node.items()}
    if isinstance(node, list):
        if len(node) == 3 and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in node):
            return [float(v) + float(rng.normal(0.0, sigma)) for v in node]
        return [perturb(v, rng, sigma) for v in node]
    return node


def main():
    from open_kinematics_amd import solver
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.solver import SolverConfig
    from open_kinematics_amd.sweep import solve_sweep

    name = sys.argv[1] if len(sys.argv) > 1 else "c1_dw_corner"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    arrays = dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz"), allow_pickle=False))
    geometry = yaml.safe_load(str(arrays["geometry_yaml"]))
    sweep_doc = yaml.safe_load(str(arrays["sweep_yaml"]))
    rng = np.random.default_rng(0)
    worst, kept, failed, built = 0.0, 0, 0, 0
    for g in range(count):
        doc = perturb(copy.deepcopy(geometry), rng, sigma) if g else geometry
        try:
            sus = build_suspension(doc)
            sweep = build_sweep(sweep_doc, sus)
        except Exception as error:  # a perturbed document the loader refuses
            print(f"geometry {g}: not built ({type(error).__name__}: {error})")
            continue
        built += 1
        try:
            slow, _ = solve_sweep(sus, sweep, SolverConfig(parallel_chains=False))
        except RuntimeError as error:
            failed += 1
            try:
                solve_sweep(sus, sweep)
                print(f"geometry {g}: the chain raised but the default did not: {error}")
            except RuntimeError as other:
                assert str(other) == str(error), (str(other), str(error))
            continue
        fast, fast_info = solve_sweep(sus, sweep)
        cold = solve_sweep(sus, sweep, SolverConfig(warm_start=False))[1]
        kept += [i.nfev for i in fast_info] == [i.nfev for i in cold]
        worst = max(worst, max(float(np.max(np.abs(a.positions[k].data - b.positions[k].data)))
                               for a, b in zip(fast, slow) for k in a.positions))
        solver.clear_program_cache()
    print(f"{name}: {built} geometries (sigma {sigma} mm), {failed} infeasible for both forms with the same message, "
          f"{kept} of {built - failed} kept their cold starts, worst |default - chain| = {worst:.3e} mm")


if __name__ == "__main__":
    main()
