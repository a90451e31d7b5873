"""
Loader of ``libokx.so`` (HIP kernels + C-ABI, ``include/okx.h``).

The product path has no CPU fallback: if the library is missing or no GPU is visible the
solve entry points raise ``RuntimeError``.  ``torch`` is imported first so that the
library binds to the same HIP runtime (same ``libamdhip64.so.7`` SONAME) that owns torch's
device allocations and streams.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

from ._abi import ABI_VERSION, Info, ProgramDesc, SolveOpts

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libokx.so")

EXPORTS = (
    "okx_abi_version",
    "okx_last_error",
    "okx_default_opts",
    "okx_device_count",
    "okx_program_create",
    "okx_program_destroy",
    "okx_solve_batch",
    "okx_eval_batch",
    "okx_rebind_design",
    "okx_program_kernel",
    "okx_program_kernel_note",
    "okx_program_shares_first_step",
    "okx_program_has_cold_body",
    "okx_program_ready",
    "okx_quad_source",
    "okx_precompile",
    "okx_tangent_batch",
    "okx_corner_metrics_batch",
    "okx_axle_metrics_batch",
    "okx_axis_rotation_batch",
    "okx_camber_shim_batch",
    "okx_expand_positions_batch",
    "okx_program_fit_predictor",
    "okx_program_has_predictor",
    "okx_program_lane_note",
    "okx_program_lane_threshold",
    "okx_program_lane_bodies",
    "okx_lane_source",
    "okx_program_enable_evaluation",
    "okx_program_evaluation",
    "okx_program_evaluation_note",
    "okx_solve_evaluated_batch",
    "okx_evaluate_batch",
    "okx_precompile_evaluation",
    "okx_plan_launch",
    "okx_program_enable_axle_evaluation",
    "okx_precompile_axle_evaluation",
    "okx_program_eval_columns",
)

# include/okx_debug.h: test hooks and profiling aids, not part of the drop-in boundary
DEBUG_EXPORTS = (
    "okx_debug_normal_equations",
    "okx_debug_quad_eval",
    "okx_debug_lane_eval",
    "okx_debug_quad_trace",
    "okx_debug_phase_profile",
    "okx_debug_plan_stats",
    "okx_debug_kernel_scratch",
    "okx_debug_lane_scratch",
)

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C open_kinematics_amd/csrc)"
        )
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; see module docstring)
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    if os.environ.get("OKX_FIT_HOST_THREADS") == "1":
        # opt-in: size torch's host thread pool to the cgroup's CPU quota.  Loading the library
        # changes nothing about the host process otherwise; bench.py and the tools call hostcpu.fit_host_threads() themselves.
        from .hostcpu import fit_host_threads

        fit_host_threads()
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    lib.okx_abi_version.restype = i32
    lib.okx_last_error.restype = C.c_char_p
    lib.okx_default_opts.argtypes = [C.POINTER(SolveOpts)]
    lib.okx_default_opts.restype = None
    lib.okx_device_count.restype = i32
    lib.okx_program_create.argtypes = [C.POINTER(ProgramDesc), C.POINTER(vp)]
    lib.okx_program_create.restype = i32
    lib.okx_program_destroy.argtypes = [vp]
    lib.okx_program_destroy.restype = None
    lib.okx_solve_batch.argtypes = [vp, C.POINTER(SolveOpts), i64, vp, vp, vp, vp, vp, vp]
    lib.okx_solve_batch.restype = i32
    lib.okx_eval_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.okx_eval_batch.restype = i32
    lib.okx_debug_normal_equations.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.okx_debug_normal_equations.restype = i32
    lib.okx_rebind_design.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.okx_rebind_design.restype = i32
    lib.okx_debug_plan_stats.argtypes = [C.POINTER(ProgramDesc), C.POINTER(i32)]
    lib.okx_debug_plan_stats.restype = i32
    lib.okx_debug_kernel_scratch.argtypes = [C.POINTER(ProgramDesc), C.POINTER(i32)]
    lib.okx_debug_kernel_scratch.restype = i32
    lib.okx_debug_lane_scratch.argtypes = [C.POINTER(ProgramDesc), C.POINTER(i32)]
    lib.okx_debug_lane_scratch.restype = i32
    lib.okx_debug_quad_trace.argtypes = [vp, vp, i64]
    lib.okx_debug_quad_trace.restype = i32
    lib.okx_debug_phase_profile.argtypes = [vp, C.POINTER(SolveOpts), i64, vp, vp, vp, vp, vp]
    lib.okx_debug_phase_profile.restype = i32
    lib.okx_program_kernel.argtypes = [vp]
    lib.okx_program_kernel.restype = C.c_char_p
    lib.okx_program_kernel_note.argtypes = [vp]
    lib.okx_program_kernel_note.restype = C.c_char_p
    lib.okx_program_shares_first_step.argtypes = [vp]
    lib.okx_program_shares_first_step.restype = i32
    lib.okx_program_has_cold_body.argtypes = [vp]
    lib.okx_program_has_cold_body.restype = i32
    lib.okx_program_ready.argtypes = [vp, i32]
    lib.okx_program_ready.restype = i32
    lib.okx_quad_source.argtypes = [C.POINTER(ProgramDesc), C.c_char_p, i64]
    lib.okx_quad_source.restype = i64
    lib.okx_precompile.argtypes = [C.POINTER(ProgramDesc)]
    lib.okx_precompile.restype = i32
    lib.okx_tangent_batch.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.okx_tangent_batch.restype = i32
    lib.okx_corner_metrics_batch.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.okx_corner_metrics_batch.restype = i32
    lib.okx_axle_metrics_batch.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.okx_axle_metrics_batch.restype = i32
    lib.okx_axis_rotation_batch.argtypes = [vp, i32, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.okx_axis_rotation_batch.restype = i32
    lib.okx_camber_shim_batch.argtypes = [vp, i64, i32, vp, vp, vp, vp]
    lib.okx_camber_shim_batch.restype = i32
    lib.okx_expand_positions_batch.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    lib.okx_expand_positions_batch.restype = i32
    lib.okx_program_fit_predictor.argtypes = [vp, vp, vp, i32, vp]
    lib.okx_program_fit_predictor.restype = i32
    lib.okx_program_has_predictor.argtypes = [vp]
    lib.okx_program_has_predictor.restype = i32
    lib.okx_debug_quad_eval.argtypes = [vp, i64, vp, vp, C.c_double, vp, vp, vp, vp, vp]
    lib.okx_debug_quad_eval.restype = i32
    lib.okx_debug_lane_eval.argtypes = [vp, i64, vp, vp, C.c_double, vp, vp, vp, vp, vp]
    lib.okx_debug_lane_eval.restype = i32
    lib.okx_program_lane_note.argtypes = [vp]
    lib.okx_program_lane_note.restype = C.c_char_p
    lib.okx_program_lane_threshold.argtypes = [vp]
    lib.okx_program_lane_threshold.restype = i64
    lib.okx_program_lane_bodies.argtypes = [vp]
    lib.okx_program_lane_bodies.restype = i32
    lib.okx_lane_source.argtypes = [C.POINTER(ProgramDesc), C.c_char_p, i64]
    lib.okx_lane_source.restype = i64
    lib.okx_program_enable_evaluation.argtypes = [vp, vp]
    lib.okx_program_enable_evaluation.restype = i32
    lib.okx_program_evaluation.argtypes = [vp]
    lib.okx_program_evaluation.restype = i32
    lib.okx_program_evaluation_note.argtypes = [vp]
    lib.okx_program_evaluation_note.restype = C.c_char_p
    lib.okx_solve_evaluated_batch.argtypes = [vp, C.POINTER(SolveOpts), i64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.okx_solve_evaluated_batch.restype = i32
    lib.okx_evaluate_batch.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.okx_evaluate_batch.restype = i32
    lib.okx_precompile_evaluation.argtypes = [C.POINTER(ProgramDesc), vp]
    lib.okx_precompile_evaluation.restype = i32
    lib.okx_plan_launch.argtypes = [vp, C.POINTER(SolveOpts), i64, i32, i32, C.POINTER(i32 * 2)]
    lib.okx_plan_launch.restype = i32
    lib.okx_program_enable_axle_evaluation.argtypes = [vp, vp]
    lib.okx_program_enable_axle_evaluation.restype = i32
    lib.okx_precompile_axle_evaluation.argtypes = [C.POINTER(ProgramDesc), vp]
    lib.okx_precompile_axle_evaluation.restype = i32
    lib.okx_program_eval_columns.argtypes = [vp]
    lib.okx_program_eval_columns.restype = i32
    if lib.okx_abi_version() != ABI_VERSION:
        raise RuntimeError("libokx.so ABI version mismatch")
    _lib = lib
    return lib


def last_error() -> str:
    return load().okx_last_error().decode("utf-8", "replace")


def device_count() -> int:
    return int(load().okx_device_count())


def check(rc: int, what: str) -> None:
    """Map okx_status to the reference's exception types (solver.py:116-121)."""
    if rc == 0:
        return
    msg = last_error()
    if rc == -3:
        raise ValueError(msg)
    if rc in (-1, -2):
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed ({rc}): {msg}")


__all__ = ["load", "build", "check", "device_count", "last_error", "Info", "SolveOpts", "LIB_PATH", "EXPORTS", "DEBUG_EXPORTS"]
