#!/usr/bin/env python3
"""Wall time of the pieces of one drop-in `solve_sweep` call (perf_counter around each, 300 repetitions; no profiler).
  python tools/dropin_phases.py [fixture] [--segment=k]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402


def main():
    from open_kinematics_amd import solver as SV
    from open_kinematics_amd import sweep as S
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.targeting import validate_sweep_controls

    args = [a for a in sys.argv[1:] if not a.startswith("--segment=")]
    forced = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--segment=")]
    if forced:  # try another chain length for the parallel chains of a warm-started sweep
        SV._segment_length = lambda n_steps: forced[0]
    name = args[0] if args else "c1_dw_corner"
    arrays = dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz"), allow_pickle=False))
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    cfg = SV.SolverConfig()
    for _ in range(5):
        S.solve_sweep(sus, sweep)
    reps = 300
    acc = {}

    def timed(label, fn):
        t0 = time.perf_counter()
        out = fn()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return out

    t_all = time.perf_counter()
    for _ in range(reps):
        S.solve_sweep(sus, sweep)
    whole = (time.perf_counter() - t_all) / reps
    for _ in range(reps):
        timed("validate_sweep_controls", lambda: validate_sweep_controls(sweep, sus.actuator_dofs()))
        timed("target_segments", lambda: SV.target_segments(sweep))
        state, (program, table) = timed("memoized_program (initial_state, table, key)", lambda: S._dropin_flattened(sus, sweep, cfg))
        dp = timed("_device_program (hash of the arrays)", lambda: SV._device_program(program, None))
        segment = SV._segment_length(table.shape[0])
        kw = dict(max_iter=cfg.max_iter, residual_tolerance=cfg.residual_tolerance, predictor=False, **SV.device_tolerances(cfg, program))
        targets = timed("as_tensor", lambda: torch.as_tensor(table))
        result = timed("dp.solve (upload + launch)", lambda: dp.solve(targets, chain_len=segment, **kw) if segment else dp.solve(targets, chain=True, **kw))
        positions, info = timed("result.host() (waits for the launch, both copies)", lambda: result.host())
        if segment:
            timed("_chains_are_continuous", lambda: SV._chains_are_continuous(program, table, positions, info, segment))
        timed("_raise_on_first_failure", lambda: SV._raise_on_first_failure(program, dp, table, positions, info, sweep, state, cfg))
        timed("_states_from_positions", lambda: SV._states_from_positions(state, program, positions))
        timed("_solver_infos", lambda: SV._solver_infos(info, dp, segment or table.shape[0]))
    print(f"{name}: {table.shape[0]} steps, solve_sweep {1e3 * whole:.3f} ms per call")
    total = 0.0
    for label, t in acc.items():
        total += t
        print(f"  {1e6 * t / reps:8.1f} us  {label}")
    print(f"  {1e6 * total / reps:8.1f} us  sum of the pieces")


if __name__ == "__main__":
    main()
