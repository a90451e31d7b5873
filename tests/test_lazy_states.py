"""
CPU: the drop-in's result containers.  ``state.RowPositions`` - the lazily materialised ``positions`` mapping of the
states ``solve_sweep`` returns - behaves as the plain dict the reference's ``SuspensionState.positions`` is
(``core/state.py:23-72``), and ``solver.memoized_program`` re-flattens when a hardpoint moves.
"""

import copy
import pickle

import numpy as np
import pytest


def _lazy():
    from open_kinematics_amd.state import Point3, RowPositions

    rows = np.arange(12.0).reshape(4, 3)
    return RowPositions(rows, {"a": 0, "b": 1, "c": 2, "d": 3}, Point3.from_trusted), rows


def test_row_positions_is_a_complete_dict():
    from open_kinematics_amd.state import Point3

    d, rows = _lazy()
    assert isinstance(d, dict) and len(d) == 4 and list(d) == ["a", "b", "c", "d"] and "c" in d and "z" not in d
    assert d.rows_if_untouched() is not None
    assert np.shares_memory(d["b"].data, rows) and d["b"] is d["b"]            # made once, a view of its row
    assert d.rows_if_untouched() is None                                      # somebody has looked: callers walk the points now
    assert dict(d).keys() == {"a", "b", "c", "d"} and {**d}.keys() == dict(d).keys()
    assert [k for k, _ in d.items()] == list(d.keys()) and len(list(d.values())) == 4
    assert d == {k: v for k, v in d.items()} and d != {"a": 1}
    d["e"] = Point3([1.0, 2.0, 3.0])
    assert len(d) == 5 and list(d)[-1] == "e"
    del d["a"]
    assert "a" not in d and len(d) == 4 and d.get("a") is None
    with pytest.raises(KeyError):
        d["a"]
    d["a"] = Point3([0.0, 0.0, 0.0])                                          # a deleted key can come back
    assert "a" in d and len(d) == 5
    assert d.pop("e").x == 1.0 and d.pop("nope", 7) == 7 and len(d) == 4
    assert d.setdefault("b", None) is d["b"]
    for clone in (copy.copy(d), copy.deepcopy(d), pickle.loads(pickle.dumps(d)), d.copy()):
        assert isinstance(clone, dict) and set(clone) == set(d)
    d.clear()
    assert len(d) == 0 and list(d) == [] and "b" not in d


def test_states_built_from_a_sweep_block_are_independent_and_lazy():
    from open_kinematics_amd.solver import _states_from_positions
    from open_kinematics_amd.state import Point3, RowPositions, SuspensionState

    class Program:  # the two attributes _states_from_positions reads
        point_keys = ["p0", "p1", "p2"]
        out_point = [0, 1, 2]

    design = SuspensionState({k: Point3([0.0, 0.0, 0.0]) for k in Program.point_keys}, {"p1", "p2"})
    block = np.arange(18.0).reshape(2, 3, 3)
    states = _states_from_positions(design, Program, block)
    block[:] = -1.0                                                            # the result owns a copy of the sweep
    assert [type(s) for s in states] == [SuspensionState, SuspensionState] and isinstance(states[0].positions, RowPositions)
    assert states[1].positions["p2"].data.tolist() == [15.0, 16.0, 17.0] and states[0].free_points_order == ["p1", "p2"]
    states[0].positions["p0"].data[0] = 99.0                                   # a write to one state leaves the other alone
    assert states[1].positions["p0"].data[0] == 9.0 and states[0].get("p0").x == 99.0
    clone = states[0].copy()
    clone.set("p1", [1.0, 1.0, 1.0])
    assert states[0]["p1"].data.tolist() == [3.0, 4.0, 5.0] and states[0].get_free_array().shape == (6,)


def test_the_program_memo_sees_a_moved_hardpoint():
    from open_kinematics_amd.input import load_geometry, load_sweep
    from open_kinematics_amd.sweep import sweep_program
    from open_kinematics_amd.workloads import geometry_path

    sus = load_geometry(geometry_path("geometry.yaml"))
    sweep = load_sweep(geometry_path("bump_sweep.yaml"), sus)
    first, table = sweep_program(sus, sweep)
    again, table2 = sweep_program(sus, sweep)
    assert again is first and np.array_equal(table, table2)                    # the same call: the memo's object
    other, _ = sweep_program(sus, sweep, line_mode="softnorm")
    assert other is not first and other.line_mode == "softnorm"
    state = sus.initial_state()
    moved = next(iter(state.free_points))
    original = sus.initial_state
    shifted = state.copy()
    shifted.positions[moved].data[0] += 1.0
    sus.initial_state = lambda: shifted                                        # the geometry under the object changes ...
    try:
        fresh, _ = sweep_program(sus, sweep)
    finally:
        sus.initial_state = original
    assert fresh is not first                                                  # ... and the program is flattened afresh
    assert not np.array_equal(fresh.design_pos, first.design_pos)
