"""
Fill the kernel cache for a user's own suspension ahead of time (no GPU needed):

    python -m open_kinematics_amd.precompile geometry.yaml sweep.yaml [more_sweeps.yaml ...]

compiles what the drop-in entry points make of the geometry - ``solve_sweep`` / ``solve_evaluated_sweep`` (every point of a
state an output), ``compute_sweep_metrics`` / ``compute_sweep_tangents`` (the suspension's own output list) and the evaluated modules of
both (a corner's, or a composed axle's pair-mode module) - so that the first sweep of a fresh process runs the generated kernels instead of the
interpreter while a compile thread works (``include/okx.h``: tiered start).  A program's target rows come from the sweep
(which actuators it drives), hence the sweep files: sweeps that drive the same actuators share their programs.  Honours
``OKX_KERNEL_CACHE``.  Exit status: the number of programs that could not be compiled.
"""

from __future__ import annotations

import ctypes as C
import sys
import time


def precompile_suspension(suspension, sweeps) -> list:
    """``[(what, status, seconds)]`` for every program of the suspension under the given sweeps."""
    from . import _lib
    from ._abi import HostProgram
    from .metrics import AxleRoles, CornerRoles, axle_evaluation_roles, corner_roles
    from .solver import dropin_program
    from .sweep import sweep_program

    lib = _lib.load()
    done, seen = [], set()
    for label, sweep in sweeps:
        pair = [("drop-in solve", dropin_program(suspension.initial_state(), suspension.constraints(), sweep, suspension.derived_spec())[0]),
                ("metrics / tangents", sweep_program(suspension, sweep)[0])]
        for what, program in pair:
            arrays = program.to_arrays()
            key = (program.line_mode, tuple((k, v.tobytes()) for k, v in sorted(arrays.items()) if k != "target_desc"))
            if key in seen:
                continue
            seen.add(key)
            host = HostProgram(program)
            t0 = time.perf_counter()
            rc = lib.okx_precompile(host.byref())
            note = "ok" if rc == 0 else ("no generated kernel for this program (the interpreter serves it)" if rc == -2 else _lib.last_error())
            if rc == 0 and not hasattr(suspension, "corners"):
                roles = corner_roles(suspension, program)
                rc = lib.okx_precompile_evaluation(host.byref(), C.byref(CornerRoles.from_buffer_copy(bytes(roles))))
                note = "ok, with the evaluated modules" if rc == 0 else "solve kernels ok; evaluated modules: " + _lib.last_error()
            elif rc == 0:  # a composed axle: the pair-mode evaluated module (both corners' roles + rotation / hardware roles)
                try:
                    roles = axle_evaluation_roles(suspension, program)[0]
                    rc = lib.okx_precompile_axle_evaluation(host.byref(), C.byref(AxleRoles.from_buffer_copy(bytes(roles))))
                    note = "ok, with the axle's evaluated module" if rc == 0 else "solve kernels ok; evaluated module: " + _lib.last_error()
                    rc = 0 if rc == -2 else rc  # (no pair-mode kernel for this axle: it evaluates in separate launches)
                except ValueError as error:
                    note = f"solve kernels ok; no evaluated module ({error})"
            done.append((f"{label}: {what}", note, time.perf_counter() - t0, rc))
    return done


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2 or argv[0] in ("-h", "--help"):
        print(__doc__.strip())
        return 0 if argv and argv[0] in ("-h", "--help") else 2
    from .input import load_geometry, load_sweep

    suspension = load_geometry(argv[0])
    sweeps = [(path, load_sweep(path, suspension)) for path in argv[1:]]
    failed = 0
    for what, note, seconds, rc in precompile_suspension(suspension, sweeps):
        print(f"{what}: {note} ({seconds:.1f} s)")
        failed += 1 if rc not in (0, -2) else 0
    return failed


if __name__ == "__main__":
    raise SystemExit(main())
