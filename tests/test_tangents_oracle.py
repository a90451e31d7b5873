"""CPU: the numpy restatement of compute_state_tangents (oracle) against the reference's own outputs."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.oracle import Oracle

FIXTURES = ["c1_dw_corner", "c4_macpherson_grid", "u_dw_corner", "u_macpherson", "c3_axle_grid"]


def load_tangent_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"tangents_{name}.npz"), allow_pickle=False))


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("mode", ["softnorm", "pinned"])
def test_oracle_tangents_match_the_reference(golden, name, mode):
    """
    softnorm = the reference's formulation (zero-gradient line row + two smooth pins,
    sensitivity.py:83-87); pinned = this build's line rows.  Both give the reference's
    velocities: the pins span the same plane with the same Gram matrix.
    """
    _, program = golden(name)
    program = program.with_line_mode(mode)
    tg = load_tangent_golden(name)
    orc = Oracle(program)
    free_out = [list(program.out_point).index(int(p)) for p in program.free_point]
    for k in range(len(tg["step_index"])):
        vel, rank, sv = orc.tangents(tg["pos"][k][free_out].reshape(-1))
        assert np.max(np.abs(vel[:, program.out_point] - tg["vel"][k])) <= 1e-12
        if mode == "softnorm":
            assert rank == tg["rank"][k] == program.n_vars
            assert sv[-1] == pytest.approx(tg["smallest_sv"][k], rel=1e-9)


def test_bump_tangent_moves_the_wheel_centre_at_unit_rate(golden):
    """tests/test_sensitivity.py:86: d wheel_center.z / d bump target = 1."""
    _, program = golden("c1_dw_corner")
    tg = load_tangent_golden("c1_dw_corner")
    wc = [i for i, k in enumerate(program.out_point) if program.point_keys[k].lower_name == "wheel_center"][0]
    assert np.max(np.abs(tg["vel"][:, 1, wc, 2] - 1.0)) <= 1e-12
    assert np.max(np.abs(tg["vel"][:, 0, wc, 2])) <= 1e-12
