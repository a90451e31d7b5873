"""
open_kinematics_amd — MI355X-native batched solver for the per-sweep-step constraint solve
of nickmccleery/open-kinematics (``kinematics.core.solver`` / ``kinematics.core.sweep``).

The hot path runs in hand-written HIP (``csrc/``) behind the C-ABI in ``include/okx.h``;
this package is the Python host side that mirrors the reference's ``solve_sweep`` /
``solve_suspension_sweep`` API.  There is no CPU fallback: without the built HIP library and
a GPU the solve entry points raise.
"""

from .program import ConstraintProgram, flatten_problem  # noqa: F401

__version__ = "0.1.0"
