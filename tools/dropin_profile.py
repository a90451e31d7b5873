#!/usr/bin/env python3
"""cProfile of the drop-in's sweep calls on one fixture (host time: the GPU work is tens of microseconds).
  python tools/dropin_profile.py c1_dw_corner [solve_sweep|compute_sweep_metrics|solve_evaluated_sweep]"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import yaml  # noqa: E402


def main():
    from open_kinematics_amd import sweep as S
    from open_kinematics_amd.input import build_suspension, build_sweep

    name = sys.argv[1] if len(sys.argv) > 1 else "c1_dw_corner"
    which = sys.argv[2] if len(sys.argv) > 2 else "solve_sweep"
    arrays = dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz"), allow_pickle=False))
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    states, stats = S.solve_sweep(sus, sweep)
    fn = {"solve_sweep": lambda: S.solve_sweep(sus, sweep), "compute_sweep_metrics": lambda: S.compute_sweep_metrics(sus, sweep, states),
          "solve_evaluated_sweep": lambda: S.solve_evaluated_sweep(sus, sweep)}[which]
    for _ in range(3):
        fn()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(20):
        fn()
    prof.disable()
    st = pstats.Stats(prof)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
