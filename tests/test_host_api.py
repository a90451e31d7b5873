"""Host-side mirror of the reference API: loader, problem emission, targets, error contract."""

import os

import numpy as np
import pytest
import yaml

from conftest import GOLDEN, STEERED, UNSTEERED
from open_kinematics_amd import input as okin
from open_kinematics_amd.enums import Axis, PointID, PointRef, Side
from open_kinematics_amd.program import flatten_problem
from open_kinematics_amd.solver import absolute_target_table, convert_targets_to_absolute
from open_kinematics_amd.state import Point3, SuspensionState
from open_kinematics_amd.sweep import sweep_program
from open_kinematics_amd.targeting import PointTarget, PointTargetAxis, SweepConfig, validate_sweep_controls

GEOM = os.path.join(GOLDEN, "geometry")


@pytest.mark.parametrize("name", [n for n in STEERED + UNSTEERED])
def test_own_loader_emits_the_reference_problem_bitwise(golden, name):
    """rows, order, point indices and design parameters are identical to
    what the reference's build_suspension/constraints()/initial_state() produced."""
    arrays, ref_program = golden(name)
    sus = okin.build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = okin.build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    heads, table = absolute_target_table(sweep, sus.initial_state())
    mine = flatten_problem(sus.initial_state(), sus.constraints(), sus.derived_spec(), heads, sus.output_points())
    a, b = mine.to_arrays(), ref_program.to_arrays()
    for key in a:
        if key == "target_desc":
            continue
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(table, arrays["targets_abs"])  # np.linspace + design projection, bitwise


def test_yaml_files_load_and_describe_the_baseline_shapes():
    dw = okin.load_geometry(os.path.join(GEOM, "geometry.yaml"))
    assert [p.name for p in dw.initial_state().free_points_order] == [
        "LOWER_WISHBONE_OUTBOARD", "UPPER_WISHBONE_OUTBOARD", "TRACKROD_INBOARD",
        "TRACKROD_OUTBOARD", "AXLE_INBOARD", "AXLE_OUTBOARD"]
    assert len(dw.constraints()) == 17 and len(dw.output_points()) == 15
    sweep = okin.load_sweep(os.path.join(GEOM, "bump_sweep.yaml"), dw)
    assert sweep.n_steps == 36 and len(sweep.target_sweeps) == 2
    axle = okin.load_geometry(os.path.join(GEOM, "axle_geometry_rocker.yaml"))
    assert len(axle.initial_state().positions) == 44 and len(axle.constraints()) == 65
    assert len(axle.output_points()) == 38
    assert axle.initial_state().free_points_order[0] == PointRef(Side.LEFT, PointID.LOWER_WISHBONE_OUTBOARD)
    right = axle.initial_state().positions[PointRef(Side.RIGHT, PointID.AXLE_OUTBOARD)]
    assert right.y == -950.0  # mirrored through Y = 0 (build.py:344-354)
    mac = okin.load_geometry(os.path.join(GEOM, "macpherson_geometry.yaml"))
    assert len(mac.constraints()) == 15 and PointID.STRUT_BOTTOM in mac.derived_spec().functions
    program, table = sweep_program(dw, sweep)
    assert program.line_mode == "pinned" and program.n_rows == 19 and table.shape == (36, 2)


def test_loader_validation_errors():
    base = yaml.safe_load(open(os.path.join(GEOM, "geometry.yaml")))
    bad = dict(base, type="five_link")
    with pytest.raises(ValueError, match="Unsupported geometry type"):
        okin.build_suspension(bad)
    missing = yaml.safe_load(open(os.path.join(GEOM, "geometry.yaml")))
    del missing["hardpoints"]["axle_inboard"]
    with pytest.raises(ValueError, match="Missing required hardpoints"):
        okin.build_suspension(missing)
    flipped = yaml.safe_load(open(os.path.join(GEOM, "geometry.yaml")))
    flipped["hardpoints"]["axle_outboard"]["y"] = -950
    with pytest.raises(ValueError, match="AXLE_OUTBOARD Y > 0"):
        okin.build_suspension(flipped)
    unsteered = yaml.safe_load(open(os.path.join(GEOM, "geometry.yaml")))
    unsteered["config"]["steering"] = {"type": "none"}
    with pytest.raises(ValueError, match="Missing required hardpoints:.*TOE_LINK"):
        okin.build_suspension(unsteered)
    with pytest.raises(ValueError, match="unexpected keys"):
        okin.build_sweep({"version": 1, "steps": 3, "targets": [], "bogus": 1})
    with pytest.raises(ValueError, match="Unsupported sweep version"):
        okin.build_sweep({"version": 2, "targets": []})


def test_sweep_control_validation():
    """targeting.py:168-186: a rack-steered corner needs exactly one rack target per step."""
    dw = okin.load_geometry(os.path.join(GEOM, "geometry.yaml"))
    only_bump = {"version": 1, "steps": 3, "targets": [
        {"point": "wheel_center", "direction": {"axis": "z"}, "start": -10, "stop": 10}]}
    with pytest.raises(ValueError, match="exactly one target for actuator 'steering rack'"):
        okin.build_sweep(only_bump, dw)
    doubled = {"version": 1, "steps": 3, "targets": [
        {"point": "trackrod_inboard", "direction": {"axis": "y"}, "start": 0, "stop": 0},
        {"point": "trackrod_inboard", "direction": {"vector": [0, 2, 0]}, "start": 0, "stop": 0}]}
    with pytest.raises(ValueError, match="found 2 at step 0"):
        okin.build_sweep(doubled, dw)
    fixed = {"version": 1, "steps": 3, "targets": [
        {"point": "lower_wishbone_inboard_front", "direction": {"axis": "z"}, "start": 0, "stop": 1}]}
    with pytest.raises(ValueError, match="is fixed"):
        okin.build_sweep(fixed, dw)
    with pytest.raises(ValueError, match="same length"):
        SweepConfig([[PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), 0.0)] * 2,
                     [PointTarget(PointID.TRACKROD_INBOARD, PointTargetAxis(Axis.Y), 0.0)] * 3])


def test_absolute_target_conversion_and_modes():
    dw = okin.load_geometry(os.path.join(GEOM, "geometry.yaml"))
    state = dw.initial_state()
    from open_kinematics_amd.enums import TargetPositionMode as M

    rel = PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), 12.5)
    absolute = PointTarget(PointID.WHEEL_CENTER, PointTargetAxis(Axis.Z), 300.0, M.ABSOLUTE)
    out = convert_targets_to_absolute([rel, absolute], state)
    assert out[0].value == state.positions[PointID.WHEEL_CENTER].z + 12.5 and out[0].mode == M.ABSOLUTE
    assert out[1] is absolute
    cfg = SweepConfig([[rel, absolute]])
    heads, table = absolute_target_table(cfg, state)
    assert table.shape == (2, 1) and table[1, 0] == 300.0


def test_state_container_semantics():
    """state.py:46-126: sorted variable order, flat pack/unpack, independent copies."""
    s = SuspensionState({PointID.AXLE_OUTBOARD: Point3([1, 2, 3]), PointID.AXLE_INBOARD: Point3([4, 5, 6]),
                         PointID.STRUT_TOP: Point3([7, 8, 9])}, {PointID.AXLE_OUTBOARD, PointID.AXLE_INBOARD})
    assert s.free_points_order == [PointID.AXLE_INBOARD, PointID.AXLE_OUTBOARD]
    assert s.fixed_points == {PointID.STRUT_TOP}
    assert np.array_equal(s.get_free_array(), [4, 5, 6, 1, 2, 3])
    c = s.copy()
    c.update_from_array(np.arange(6, dtype=float))
    assert s[PointID.AXLE_INBOARD].x == 4.0 and c[PointID.AXLE_INBOARD].x == 0.0
    with pytest.raises(ValueError):
        c.update_from_array(np.zeros(5))


def test_flatten_rejects_unknown_pieces():
    dw = okin.load_geometry(os.path.join(GEOM, "geometry.yaml"))

    class Mystery:
        involved_points = set()

    with pytest.raises(TypeError, match="No device implementation"):
        flatten_problem(dw.initial_state(), [Mystery()], dw.derived_spec())
    with pytest.raises(ValueError, match="line_mode"):
        flatten_problem(dw.initial_state(), dw.constraints(), dw.derived_spec(), line_mode="magic")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/kinematics"),
                    reason="the reference only exists in the build container")
def test_reference_objects_pass_through_the_drop_in_front_end(golden):
    """the reference's own Suspension / SweepConfig objects are accepted as they
    are (duck typing) and flatten to the same program as the committed golden."""
    from oracle import ref_shim

    ref_shim.install()
    from kinematics.core.input import build_suspension, build_sweep  # the REAL reference

    for name in ("c1_dw_corner", "c3_axle_grid", "c4_macpherson_grid"):
        arrays, ref_program = golden(name)
        sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
        sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
        program, table = sweep_program(sus, sweep, line_mode="softnorm")
        a, b = program.to_arrays(), ref_program.to_arrays()
        for key in a:
            if key not in ("target_desc", "point_names"):
                assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(table, arrays["targets_abs"])
        # the reference's own ActuatorDOF objects drive the control validation
        validate_sweep_controls(sweep, sus.actuator_dofs())


def test_parallel_chain_policy_and_continuity_check():
    """solver._segment_length / _chains_are_continuous on synthetic paths (the device part is tests/test_gpu_dropin.py)."""
    from types import SimpleNamespace

    from open_kinematics_amd import solver

    assert [solver._segment_length(n) for n in (1, 3, 4, 36, 101, 1000)] == [0, 0, 1, 1, 1, 1]
    program = SimpleNamespace(out_point=np.arange(4), free_point=np.array([1, 3]))
    steps = 40
    table = np.stack([np.linspace(0.0, 39.0, steps), np.zeros(steps)], axis=1)
    s = table[:, 0]
    pos = np.zeros((steps, 4, 3))
    pos[:, 1, 0] = 2.0 * s + 0.01 * s * s   # smooth path of the first free point
    pos[:, 3, 2] = -s
    info = np.zeros(steps, dtype=[("flags", "<i4")])
    info["flags"] = 1
    assert solver._chains_are_continuous(program, table, pos, info, 8)
    other_branch = pos.copy()
    other_branch[16:24, 3, 2] += 3.0            # one chain landed three steps away from the path
    assert not solver._chains_are_continuous(program, table, other_branch, info, 8)
    flagged = info.copy()
    flagged["flags"][5] = 0
    assert not solver._chains_are_continuous(program, table, pos, flagged, 8)
    # chains of one step (every state a cold start): the second state is held against the two states after it
    assert solver._chains_are_continuous(program, table, pos, info, 1)
    for k in (1, 2, 17, steps - 1):
        off = pos.copy()
        off[k, 3, 2] += 3.0
        assert not solver._chains_are_continuous(program, table, off, info, 1), k
    # everything after the first state on another branch, behind a first target increment ten times the others (the
    # secant through states 0 and 1 is then too long to notice): the first state, held against the two after it, does
    uneven = table.copy()
    uneven[1:, 0] += 9.0
    s2 = uneven[:, 0]
    path = np.zeros((steps, 4, 3))
    path[:, 1, 0] = 2.0 * s2
    path[:, 3, 2] = -s2
    assert solver._chains_are_continuous(program, uneven, path, info, 1)
    away = path.copy()
    away[1:, 3, 2] += 30.0   # (the tolerance is half the extrapolated step: 10 here)
    assert not solver._chains_are_continuous(program, uneven, away, info, 1)
    assert not solver._chains_are_continuous(program, table[:3], pos[:3], info[:3], 1)   # nothing to hold state 1 against
    assert solver._chains_are_continuous(program, table[:4], pos[:4], info[:4], 1)
    missing = SimpleNamespace(out_point=np.arange(3), free_point=np.array([1, 3]))  # a free point that is not an output
    assert not solver._chains_are_continuous(missing, table, pos[:, :3], info, 8)


def test_bench_helpers():
    import bench
    from types import SimpleNamespace

    cores, how = bench.host_cores()
    assert cores >= 1 and ("affinity" in how or "quota" in how)
    dw = SimpleNamespace(n_targets=2, n_out=15)
    assert bench.algorithmic_bytes_per_solve(dw) == 392.0
    assert abs(bench.algorithmic_bytes_per_solve(dw, 256) - 393.46875) < 1e-9   # C5: + (10 x 24 + 17 x 8) / 256
    assert bench.algorithmic_bytes_per_solve(SimpleNamespace(n_targets=3, n_out=38)) == 952.0
    assert bench.algorithmic_bytes_per_solve(SimpleNamespace(n_targets=2, n_out=14)) == 368.0
    assert bench.committed_traffic("no_such_tag") == (None, None)


def test_host_thread_pool_is_fitted_to_a_cgroup_quota(monkeypatch, tmp_path):
    """hostcpu.fit_host_threads: torch's intra-op pool is cut to the cgroup's CPU quota (shared between the ranks of a
    node), never raised, left alone without a binding quota or with OKX_KEEP_HOST_THREADS=1; the first call decides."""
    import builtins

    import torch

    from open_kinematics_amd import hostcpu

    real_open = builtins.open
    quota = {"text": "400000 100000"}

    def fake_open(path, *a, **kw):
        if path == "/sys/fs/cgroup/cpu.max":
            fake = tmp_path / "cpu.max"
            fake.write_text(quota["text"])
            return real_open(fake, *a, **kw)
        return real_open(path, *a, **kw)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(hostcpu.os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert hostcpu.host_cores() == (4, "cgroup CPU quota of 4 inside an affinity mask of 64")
    quota["text"] = "max 100000"
    assert hostcpu.host_cores() == (64, "affinity mask of 64")
    quota["text"] = "1600000 100000"
    before = torch.get_num_threads()
    calls = []
    monkeypatch.setattr(torch, "get_num_threads", lambda: 128)
    monkeypatch.setattr(torch, "set_num_threads", calls.append)
    try:
        monkeypatch.setattr(hostcpu, "_fitted", None)
        monkeypatch.setenv("OKX_KEEP_HOST_THREADS", "1")
        assert hostcpu.fit_host_threads()["changed"] is False and calls == []
        monkeypatch.delenv("OKX_KEEP_HOST_THREADS")
        monkeypatch.setattr(hostcpu, "_fitted", None)
        did = hostcpu.fit_host_threads()
        assert did["changed"] and did["host_cores"] == 16 and calls == [14]       # the quota minus two
        assert hostcpu.fit_host_threads(processes=8) is did and calls == [14]     # the first call decides
        monkeypatch.setattr(hostcpu, "_fitted", None)
        assert hostcpu.fit_host_threads(processes=4)["changed"] and calls[-1] == 2   # four ranks share the quota
        monkeypatch.setattr(hostcpu, "_fitted", None)
        quota["text"] = "max 100000"
        assert hostcpu.fit_host_threads()["changed"] is False                      # an affinity mask alone sized the pool already
    finally:
        monkeypatch.setattr(hostcpu, "_fitted", None)
    assert torch.get_num_threads() in (before, 128)


def test_precompile_entry_point_for_a_users_geometry(capsys):
    """python -m open_kinematics_amd.precompile geometry.yaml sweep.yaml: the drop-in's programs of a suspension into the
    kernel cache (the packaged fixtures' are there already: nothing is compiled here), usage text without arguments."""
    from open_kinematics_amd import precompile
    from open_kinematics_amd.workloads import geometry_path

    assert precompile.main([]) == 2 and "geometry.yaml sweep.yaml" in capsys.readouterr().out
    assert precompile.main([geometry_path("geometry.yaml"), geometry_path("sweep.yaml"), geometry_path("bump_sweep.yaml")]) == 0
    out = capsys.readouterr().out
    assert "drop-in solve: ok, with the evaluated modules" in out
    assert precompile.main([geometry_path("axle_geometry.yaml"), geometry_path("axle_sweep.yaml")]) == 0
    out = capsys.readouterr().out
    assert "drop-in solve: ok, with the axle's evaluated module" in out   # axles: the pair-mode solve kernels and (round 6) their evaluated module
