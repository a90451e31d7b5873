"""The drop-in boundary on the GPU: solve_sweep / solve_suspension_sweep with the reference's
call shapes, result types and error contract (tests/core/test_solver.py)."""

import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GEOM = os.path.join(GOLDEN, "geometry")


def _dw():
    from open_kinematics_amd.input import load_geometry

    return load_geometry(os.path.join(GEOM, "geometry.yaml"))


def test_solve_sweep_returns_reference_shaped_results(golden):
    from open_kinematics_amd.input import build_sweep
    from open_kinematics_amd.solver import SolverInfo
    from open_kinematics_amd.state import SuspensionState
    from open_kinematics_amd.sweep import solve_sweep
    import yaml

    arrays, program = golden("c1_dw_corner")
    sus = _dw()
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    initial_before = sus.initial_state().get_free_array().copy()
    states, infos = solve_sweep(sus, sweep)
    assert len(states) == len(infos) == 101
    assert all(isinstance(s, SuspensionState) for s in states) and isinstance(infos[0], SolverInfo)
    assert all(i.converged and i.max_residual < 1e-3 and i.nfev >= 2 for i in infos)
    # every state holds ALL points (fixed + free + derived) as independent copies
    assert set(states[0].positions) == set(sus.initial_state().positions)
    assert states[0].positions is not states[1].positions
    assert np.array_equal(sus.initial_state().get_free_array(), initial_before)  # inputs untouched
    out = sus.output_points()
    pos = np.array([[s.positions[k].data for k in out] for s in states])
    diff = np.abs(pos - arrays["ref_tight_pos"])
    assert diff.max() <= 6e-8 and np.abs(pos - arrays["ref_default_pos"]).max() <= 5e-5
    # the sequential warm start (reference semantics) needs fewer evaluations than independent cold starts - which is what
    # the default solves the sweep as, side by side, before checking that they are the sequential path
    from open_kinematics_amd.solver import SolverConfig

    _, cold = solve_sweep(sus, sweep, SolverConfig(warm_start=False))
    _, chain = solve_sweep(sus, sweep, SolverConfig(parallel_chains=False))
    assert sum(i.nfev for i in chain) < sum(i.nfev for i in cold) == sum(i.nfev for i in infos)


def test_infeasible_step_raises_like_the_reference():
    """solver.py:735-747 / tests/core/test_solver.py:210-231."""
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.sweep import solve_sweep
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig

    sus = _dw()
    rack = [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 0.0) for _ in range(3)]
    bump = [PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), v) for v in (0.0, 10.0, 1500.0)]
    with pytest.raises(RuntimeError, match=r"sweep step 2.*Worst residual row"):
        solve_sweep(sus, SweepConfig([rack, bump]))


def test_underdetermined_system_is_a_value_error():
    """solver.py:116-121 / tests/core/test_solver.py:198-207."""
    from open_kinematics_amd.constraints import DistanceConstraint
    from open_kinematics_amd.derived import DerivedPointsSpec
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.solver import solve_suspension_sweep
    from open_kinematics_amd.state import Point3, SuspensionState
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig

    P = PointID
    state = SuspensionState({P.LOWER_WISHBONE_INBOARD_FRONT: Point3([0, 0, 0]),
                             P.LOWER_WISHBONE_OUTBOARD: Point3([100, 0, 0])}, {P.LOWER_WISHBONE_OUTBOARD})
    cons = [DistanceConstraint(P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_OUTBOARD, 100.0)]
    sweep = SweepConfig([[PointTarget(P.LOWER_WISHBONE_OUTBOARD, PointTargetAxis(Axis.Z), 1.0)]])
    with pytest.raises(ValueError, match="System is underdetermined"):
        solve_suspension_sweep(state, cons, sweep, DerivedPointsSpec({}, {}))


def test_toy_two_link_sweep():
    """The reference's 1-point / 2-distance toy (tests/core/test_solver.py:143-195)."""
    from open_kinematics_amd.constraints import DistanceConstraint, FixedAxisConstraint
    from open_kinematics_amd.derived import DerivedPointsSpec
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.solver import solve_suspension_sweep
    from open_kinematics_amd.state import Point3, SuspensionState
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig

    P = PointID
    a, b, free = P.LOWER_WISHBONE_INBOARD_FRONT, P.LOWER_WISHBONE_INBOARD_REAR, P.LOWER_WISHBONE_OUTBOARD
    state = SuspensionState({a: Point3([0, 0, 0]), b: Point3([0, 200, 0]), free: Point3([300, 100, 0])}, {free})
    length = float(np.hypot(300, 100))
    cons = [DistanceConstraint(a, free, length), DistanceConstraint(b, free, length)]
    sweep = SweepConfig([[PointTarget(free, PointTargetAxis(Axis.Z), v) for v in (0.0, 25.0, 50.0)]])
    states, infos = solve_suspension_sweep(state, cons, sweep, DerivedPointsSpec({}, {}))
    for s, z in zip(states, (0.0, 25.0, 50.0)):
        p = s.positions[free].data
        assert abs(p[2] - z) < 1e-9 and abs(p[1] - 100.0) < 1e-6
        assert abs(np.linalg.norm(p) - length) < 1e-5
    assert all(i.converged for i in infos)


def test_repeated_calls_reuse_the_device_program():
    """solver._PROGRAM_CACHE: the reference's benchmark calls solve_sweep over and over on one suspension."""
    import time

    from open_kinematics_amd import solver
    from open_kinematics_amd.input import load_geometry, load_sweep
    from open_kinematics_amd.sweep import compute_sweep_metrics, solve_sweep

    solver.clear_program_cache()
    axle = load_geometry(os.path.join(GEOM, "axle_geometry_rocker.yaml"))
    sweep = load_sweep(os.path.join(GEOM, "axle_rocker_sweep.yaml"), axle)
    t0 = time.perf_counter()
    states, infos = solve_sweep(axle, sweep)
    first = time.perf_counter() - t0
    assert len(solver._PROGRAM_CACHE) == 1
    cached = next(iter(solver._PROGRAM_CACHE.values()))
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        again, _ = solve_sweep(axle, sweep)
        times.append(time.perf_counter() - t0)
    assert len(solver._PROGRAM_CACHE) == 1 and next(iter(solver._PROGRAM_CACHE.values())) is cached
    for a, b in zip(states, again):
        for key in a.positions:
            assert np.array_equal(a.positions[key].data, b.positions[key].data)
    assert again[0].positions is not states[0].positions
    assert min(times) < first  # no upload / generation / code-object load on the way
    # the metrics entry point shares the cache (its program lists Suspension.output_points(): a second entry)
    result = compute_sweep_metrics(axle, sweep, states)
    assert len(result.rows) == len(states) and len(solver._PROGRAM_CACHE) == 2
    assert result.tangent_solve_infos is not None and len(result.tangent_solve_infos) == len(states)
    assert all(not info.rank_deficient and info.n_variables == 60 for info in result.tangent_solve_infos)
    # a different geometry value is a different program
    other = load_geometry(os.path.join(GEOM, "geometry.yaml"))
    from open_kinematics_amd.input import load_sweep as load
    solve_sweep(other, load(os.path.join(GEOM, "bump_sweep.yaml"), other))
    assert len(solver._PROGRAM_CACHE) == 3
    solver.clear_program_cache()
    # the cache only forgets: a program another holder still has stays usable and is released with its last reference
    assert not solver._PROGRAM_CACHE and cached._handle is not None
    assert cached.kernel in ("quad", "wave")
    # 'cuda' and 'cuda:0' are one entry
    solve_sweep(other, load(os.path.join(GEOM, "bump_sweep.yaml"), other), device="cuda")
    solve_sweep(other, load(os.path.join(GEOM, "bump_sweep.yaml"), other), device="cuda:0")
    assert len(solver._PROGRAM_CACHE) == 1
    solver.clear_program_cache()


def test_parallel_chains_give_the_sequential_answer_or_fall_back(golden):
    """Long warm-started sweeps run as several chains at once; the result is kept only when it is the sequential path."""
    import yaml

    from open_kinematics_amd import solver
    from open_kinematics_amd.input import build_sweep
    from open_kinematics_amd.solver import SolverConfig
    from open_kinematics_amd.sweep import solve_sweep

    arrays, _ = golden("c1_dw_corner")
    sus = _dw()
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)  # 101 steps: 101 cold starts side by side
    assert solver._segment_length(101) == 1 and solver._segment_length(3) == 0
    fast, fast_info = solve_sweep(sus, sweep)
    slow, slow_info = solve_sweep(sus, sweep, SolverConfig(parallel_chains=False))
    for a, b in zip(fast, slow):
        for key in a.positions:
            assert np.max(np.abs(a.positions[key].data - b.positions[key].data)) <= 1e-9
    # every step started cold: the evaluations of 101 independent solves (so the parallel result was the one kept, not
    # the sequential fallback), more than one chain's
    cold = solve_sweep(sus, sweep, SolverConfig(warm_start=False))[1]
    assert [i.nfev for i in fast_info] == [i.nfev for i in cold]
    assert sum(i.nfev for i in slow_info) <= sum(i.nfev for i in fast_info)

    # the continuity test itself: a chain head on another branch, a rejected step and a kinked target path all fail it
    program, table = solver.dropin_program(sus.initial_state(), sus.constraints(), sweep, sus.derived_spec())
    pos = np.array([[s.positions[k].data for k in program.point_keys] for s in fast])
    info = np.zeros(101, dtype=[("flags", "<i4")])
    info["flags"] = 1
    assert solver._chains_are_continuous(program, table, pos, info, 11)
    jumped = pos.copy()
    jumped[22:33, program.free_point[0]] += 5.0
    assert not solver._chains_are_continuous(program, table, jumped, info, 11)
    bad = info.copy()
    bad["flags"][40] = 3
    assert not solver._chains_are_continuous(program, table, pos, bad, 11)
    kinked = table.copy()
    kinked[33:] = kinked[33:][::-1]
    assert not solver._chains_are_continuous(program, kinked, pos, info, 11)


@pytest.mark.parametrize("name", ["t_corner_rocker", "e2e_sweep", "t_axle_dw", "t_axle_macpherson", "t_axle_heave_link",
                                  "t_axle_t_bar_bump", "t_axle_t_bar_roll", "t_axle_t_bar_heave"])
def test_parallel_cold_starts_are_the_sequential_path_on_the_reference_fixtures(golden, name):
    """Every fixture sweep of the reference, corners and composed axles: the cold starts side by side are kept (they pass
    the continuity test) and are the sequential warm start's states to 1e-9 mm."""
    import yaml

    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.solver import SolverConfig
    from open_kinematics_amd.sweep import solve_sweep

    arrays, _ = golden(name)
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    fast, fast_info = solve_sweep(sus, sweep)
    slow, _ = solve_sweep(sus, sweep, SolverConfig(parallel_chains=False))
    cold = solve_sweep(sus, sweep, SolverConfig(warm_start=False))[1]
    assert len(fast) == len(slow) == sweep.n_steps
    if sweep.n_steps >= 4:
        assert [i.nfev for i in fast_info] == [i.nfev for i in cold], "fell back to the sequential chain"
    worst = max(float(np.max(np.abs(a.positions[key].data - b.positions[key].data))) for a, b in zip(fast, slow) for key in a.positions)
    assert worst <= 1e-9, worst


def test_infeasible_long_sweep_raises_at_the_sequential_step():
    """The parallel attempt is discarded and the reference's first-failure semantics come from the sequential chain."""
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.solver import SolverConfig
    from open_kinematics_amd.sweep import solve_sweep
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig
    from open_kinematics_amd.enums import TargetPositionMode

    sus = _dw()
    z = PointTargetAxis(Axis.Z)
    y = PointTargetAxis(Axis.Y)
    bump = np.linspace(0.0, 600.0, 64)  # runs past lock-out (bump reach ~ 451 mm)
    sweep = SweepConfig([[PointTarget(PointID.WHEEL_CENTER, z, float(v), TargetPositionMode.RELATIVE) for v in bump],
                         [PointTarget(PointID.TRACKROD_INBOARD, y, 0.0, TargetPositionMode.RELATIVE) for _ in bump]])
    messages = []
    for config in (SolverConfig(), SolverConfig(parallel_chains=False)):
        with pytest.raises(RuntimeError) as err:
            solve_sweep(sus, sweep, config)
        messages.append(str(err.value))
    assert messages[0] == messages[1] and "did not reach an acceptable residual" in messages[0]


def test_sweep_whose_target_point_changes_mid_sweep_is_solved_run_by_run():
    """The reference pairs arbitrary PointTargets by index (targeting.py:67-75): the bump dimension drives the wheel centre
    for the first steps and then the contact patch, the rack dimension rebuilds its (equal) direction object.  Every run of steps
    with one set of target rows is its own program; the next run starts where the last one ended.  Checked against the
    CPU oracle solving the same runs (MINPACK to 1e-15 on the same rows, warm-started the same way)."""
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.program import flatten_problem
    from open_kinematics_amd.solver import SolverConfig, absolute_target_table, solve_suspension_sweep, target_segments
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig
    from oracle.oracle import Oracle

    sus = _dw()
    n = 12
    z = PointTargetAxis(Axis.Z)
    bump = [PointTarget(PointID.WHEEL_CENTER, z, 4.0 * k) for k in range(5)] + \
           [PointTarget(PointID.CONTACT_PATCH_CENTER, z, 16.0 + 3.0 * k) for k in range(1, n - 4)]
    rack = [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 0.5 * k) for k in range(8)] + \
           [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 3.5 - 0.25 * k) for k in range(1, n - 7)]
    sweep = SweepConfig([rack, bump])
    assert target_segments(sweep) == [(0, 5), (5, n)]   # the rack dimension keeps point and direction value throughout
    with pytest.raises(NotImplementedError, match="run by run"):
        absolute_target_table(sweep, sus.initial_state())
    states, infos = solve_suspension_sweep(sus.initial_state(), sus.constraints(), sweep, sus.derived_spec())
    assert len(states) == len(infos) == n and all(i.converged and i.max_residual < 1e-3 for i in infos)
    # the oracle on the same two runs, the second started from the first's last state
    initial = sus.initial_state()
    start = initial
    for lo, hi in target_segments(sweep):
        heads, table = absolute_target_table(sweep, initial, (lo, hi))
        program = flatten_problem(start, sus.constraints(), sus.derived_spec(), heads, None, line_mode="softnorm").with_line_mode("pinned")
        orc = Oracle(program).sweep(table, 1e-15, 1e-15, 1e-15, warm_start=True)
        assert orc.first_failed_step == -1
        keys = [program.point_keys[k] for k in program.out_point]
        mine = np.array([[states[s].positions[k].data for k in keys] for s in range(lo, hi)])
        assert np.max(np.abs(mine - orc.positions)) <= 1e-9
        start = states[hi - 1]
    # the targets are met: wheel centre z in the first run, contact patch z in the second (relative to the design state)
    wc0 = initial.positions[PointID.WHEEL_CENTER].data[2]
    cp0 = initial.positions[PointID.CONTACT_PATCH_CENTER].data[2]
    assert abs(states[4].positions[PointID.WHEEL_CENTER].data[2] - (wc0 + 16.0)) <= 1e-9
    assert abs(states[n - 1].positions[PointID.CONTACT_PATCH_CENTER].data[2] - (cp0 + 16.0 + 3.0 * (n - 5))) <= 1e-9
    # an infeasible step inside the second run is reported with its index in the whole sweep
    bad = list(bump)
    bad[8] = PointTarget(PointID.CONTACT_PATCH_CENTER, z, 1500.0)
    with pytest.raises(RuntimeError, match=r"sweep step 8.*Worst residual row"):
        solve_suspension_sweep(sus.initial_state(), sus.constraints(), SweepConfig([rack, bad]), sus.derived_spec())


def test_restricted_output_points_and_an_unreachable_target_still_raise_the_references_error():
    """the failure path indexed the output-ordered record with program point indices."""
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.solver import solve_suspension_sweep
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig

    sus = _dw()
    rack = [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 0.0) for _ in range(3)]
    bump = [PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), v) for v in (0.0, 10.0, 1500.0)]
    few = [PointID.WHEEL_CENTER, PointID.CONTACT_PATCH_CENTER]
    with pytest.raises(RuntimeError, match=r"sweep step 2 did not reach an acceptable residual.*Worst residual row"):
        solve_suspension_sweep(sus.initial_state(), sus.constraints(), SweepConfig([rack, bump]), sus.derived_spec(),
                               output_points=few)
    ok = [PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), v) for v in (0.0, 10.0, 20.0)]
    states, infos = solve_suspension_sweep(sus.initial_state(), sus.constraints(), SweepConfig([rack, ok]), sus.derived_spec(),
                                           output_points=few)
    assert [set(s.positions) for s in states] == [set(few)] * 3


def test_solver_info_counts_the_shared_design_state_evaluation_for_chain_heads():
    """The device's nfev counts what a problem ran itself; the drop-in adds the evaluation the shared first step made on
    a chain head's behalf (the reference counts it: solver.py:766-771)."""
    from open_kinematics_amd.enums import Axis, PointID
    from open_kinematics_amd.solver import SolverConfig, solve_suspension_sweep
    from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig

    sus = _dw()
    rack = [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 0.0) for _ in range(6)]
    bump = [PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), 5.0 * k) for k in range(6)]
    from open_kinematics_amd import solver

    solver.clear_program_cache()
    sweep = SweepConfig([rack, bump])
    _, cold = solve_suspension_sweep(sus.initial_state(), sus.constraints(), sweep, sus.derived_spec(), SolverConfig(warm_start=False))
    _, warm = solve_suspension_sweep(sus.initial_state(), sus.constraints(), sweep, sus.derived_spec(), SolverConfig(parallel_chains=False))
    _, default = solve_suspension_sweep(sus.initial_state(), sus.constraints(), sweep, sus.derived_spec())
    assert [i.nfev for i in default] == [i.nfev for i in cold]  # six steps: solved as cold starts side by side, and kept
    dp = next(iter(solver._PROGRAM_CACHE.values()))
    assert dp.shares_first_step
    _, table = solver.absolute_target_table(sweep, sus.initial_state())
    raw_cold = dp.solve(torch.as_tensor(table), chain_len=1, predictor=False).info()["nfev"]
    raw_warm = dp.solve(torch.as_tensor(table), chain=True, predictor=False).info()["nfev"]
    assert [i.nfev for i in cold] == (raw_cold + 1).tolist()                      # every cold start is a chain head
    assert [i.nfev for i in warm] == [int(raw_warm[0]) + 1] + raw_warm[1:].tolist()  # one chain: its head alone
    own = dp.solve(torch.as_tensor(table), chain_len=1, predictor=False, shared_first_step=False).info()["nfev"]
    assert np.all(np.abs(own - (raw_cold + 1)) <= 1)   # the same count as a problem that evaluates the design state itself
    solver.clear_program_cache()


def test_loosened_minpack_tolerances_are_honoured(golden):
    """SolverConfig.xtol / ftol beyond the reference's defaults (solver.py:65-80, :158-169) reach the device as its step and
    cost tolerances: the sweep stops earlier (fewer evaluations) and stays inside the reference's own default-tolerance band
    (5e-5 mm); the defaults and tighter values keep the fixed-point stop; a loosened gtol becomes the device's scaled gradient test."""
    import warnings

    import yaml

    from open_kinematics_amd.input import build_sweep
    from open_kinematics_amd.solver import SolverConfig, device_tolerances
    from open_kinematics_amd.sweep import solve_sweep

    arrays, program = golden("c1_dw_corner")
    sus = _dw()
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    out = sus.output_points()

    def run(cfg):
        states, infos = solve_sweep(sus, sweep, cfg)
        return np.array([[s.positions[k].data for k in out] for s in states]), sum(i.nfev for i in infos)

    tight_pos, tight_nfev = run(SolverConfig(warm_start=False))
    loose_pos, loose_nfev = run(SolverConfig(warm_start=False, xtol=1e-8, ftol=1e-4))  # steps below ~1e-5 mm end the solve
    assert loose_nfev < tight_nfev
    assert np.abs(loose_pos - arrays["ref_tight_pos"]).max() <= 5e-5
    assert np.abs(loose_pos - tight_pos).max() > 0.0            # it really stopped somewhere else
    same_pos, same_nfev = run(SolverConfig(warm_start=False, ftol=1e-15, xtol=1e-15, gtol=1e-15))  # the "tight" rung: nothing to loosen
    assert same_nfev == tight_nfev and np.array_equal(same_pos, tight_pos)
    norm = float(np.linalg.norm(program.design_pos[program.free_point]))
    assert device_tolerances(SolverConfig(), program) == {"step_tol": 1e-11}
    loose = device_tolerances(SolverConfig(xtol=1e-8, ftol=1e-4), program)
    assert loose["ftol"] == 1e-4 and abs(loose["step_tol"] - 1e-8 * norm) <= 1e-12 * norm
    with warnings.catch_warnings(record=True) as caught:   # a loosened gtol is MINPACK's scaled gradient test on the device: no warning
        warnings.simplefilter("always")
        assert device_tolerances(SolverConfig(gtol=1e-3), program) == {"step_tol": 1e-11, "grad_tol": -1e-3}
    assert not caught
    gt_pos, gt_nfev = run(SolverConfig(warm_start=False, gtol=1e-3))   # inside the reach the residual vanishes: the test never fires
    assert np.abs(gt_pos - tight_pos).max() <= 1e-9
