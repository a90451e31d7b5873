import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    """(arrays, softnorm ConstraintProgram) of one committed fixture."""
    from open_kinematics_amd.program import ConstraintProgram

    arrays = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    program = ConstraintProgram.from_arrays(arrays, prefix="prog_")
    return arrays, program


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name: str):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


BASELINE_STEERED = ["c1_dw_corner", "c2_dw_subset", "c3_axle_grid", "c4_macpherson_grid", "e2e_sweep"]
# every other reference geometry the loader accepts (oracle/gen_golden_topologies.py): plain and explicit double-wishbone
# axles, the MacPherson axle, corners with pushrod + rocker + torsion bar / coil-over, the rigid T-bar anti-roll bar
# (in phase / opposed wheel travel), a rocker-to-rocker heave link on the U-bar axle (66 variables, pair mode with 11
# free points per half) and on the T-bar axle (66 variables, three joining rows: two wavefronts per problem)
TOPOLOGIES = ["t_axle_dw", "t_axle_dw_explicit", "t_axle_macpherson", "t_corner_rocker", "t_corner_strut",
              "t_corner_strut_rocker", "t_axle_t_bar_bump", "t_axle_t_bar_roll", "t_axle_heave_link", "t_axle_t_bar_heave"]
STEERED = BASELINE_STEERED + TOPOLOGIES
UNSTEERED = ["u_dw_corner", "u_macpherson", "u_axle"]
ALL_ROW_CLASSES = ["rows_all_classes"]  # synthetic: one row of each of the reference's 13 constraint classes


def gpu_available() -> bool:
    try:
        from open_kinematics_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False
