"""ctypes mirror of ``include/okx.h`` (struct layouts and the program descriptor)."""

from __future__ import annotations

import ctypes as C

import numpy as np

from .program import ConstraintProgram

ABI_VERSION = 6

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class ProgramDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_points", C.c_int32),
        ("n_free", C.c_int32),
        ("n_derived", C.c_int32),
        ("n_rows", C.c_int32),
        ("n_targets", C.c_int32),
        ("n_out", C.c_int32),
        ("reserved", C.c_int32),
        ("free_point", _i32p),
        ("dop_type", _i32p),
        ("dop_out", _i32p),
        ("dop_pts", _i32p),
        ("dop_param", _f64p),
        ("row_type", _i32p),
        ("row_pts", _i32p),
        ("row_param", _f64p),
        ("tgt_point", _i32p),
        ("tgt_dir", _f64p),
        ("out_point", _i32p),
        ("design_pos", _f64p),
    ]


class SolveOpts(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32),
        ("chain", C.c_int32),
        ("steps_per_geometry", C.c_int64),
        ("chain_len", C.c_int64),
        ("step_tol", C.c_double),
        ("grad_tol", C.c_double),
        ("ftol", C.c_double),
        ("lambda0", C.c_double),
        ("residual_tolerance", C.c_double),
        ("kernel", C.c_int32),
        ("confirm_full_pass", C.c_int32),
        ("predictor", C.c_int32),
        ("shared_first_step", C.c_int32),
        ("output", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Info(C.Structure):
    _fields_ = [
        ("max_residual", C.c_double),
        ("cost", C.c_double),
        ("last_step", C.c_double),
        ("iterations", C.c_int32),
        ("nfev", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32),
    ]


INFO_DTYPE = np.dtype(
    [
        ("max_residual", "<f8"),
        ("cost", "<f8"),
        ("last_step", "<f8"),
        ("iterations", "<i4"),
        ("nfev", "<i4"),
        ("flags", "<i4"),
        ("reserved", "<i4"),
    ]
)
assert INFO_DTYPE.itemsize == C.sizeof(Info) == 40

TANGENT_INFO_DTYPE = np.dtype([("min_pivot", "<f8"), ("max_pivot", "<f8"), ("flags", "<i4"), ("reserved", "<i4")])
assert TANGENT_INFO_DTYPE.itemsize == 24
TANGENT_OK = 1
TANGENT_RANK_DEFICIENT = 2

# okx_solve_evaluated_batch / okx_evaluate_batch: d_eval [B][1 + T][EVAL_COLUMNS] (include/okx.h OKX_EVAL_*)
EVAL_COLUMNS = 24
EVAL_MIN_PIVOT, EVAL_MAX_PIVOT, EVAL_TANGENT_FLAGS = 19, 20, 21                      # row 0
EVAL_RATE_WHEEL_CENTER_X, EVAL_RATE_WHEEL_CENTER_Z, EVAL_RATE_RACK_Y = 19, 21, 22    # rows 1 + t
# a composed axle's rows (okx_program_enable_axle_evaluation): [left corner block | right corner block | axle metrics | roles]
EVAL_AXLE_COLUMNS = 64
EVAL_AXLE_LEFT, EVAL_AXLE_RIGHT, EVAL_AXLE_METRICS, EVAL_AXLE_ROLES = 0, 24, 48, 56

INFO_CONVERGED = 1
INFO_RESIDUAL_EXCEEDED = 2
INFO_FAILED = 4
INFO_ILL_CONDITIONED = 8  # advisory: pivot ratio of the last factorisation below 1e-12 (cond(J) above ~1e6)


class HostProgram:
    """Keeps the numpy buffers behind a ``ProgramDesc`` alive."""

    def __init__(self, program: ConstraintProgram):
        program.validate()
        self.program = program
        self._keep = []

        def i32(a) -> _i32p:
            arr = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
            if arr.size == 0:
                arr = np.zeros(1, dtype=np.int32)
            self._keep.append(arr)
            return arr.ctypes.data_as(_i32p)

        def f64(a) -> _f64p:
            arr = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
            if arr.size == 0:
                arr = np.zeros(1, dtype=np.float64)
            self._keep.append(arr)
            return arr.ctypes.data_as(_f64p)

        p = program
        self.desc = ProgramDesc(
            abi_version=ABI_VERSION,
            n_points=p.n_points,
            n_free=p.n_free,
            n_derived=p.n_derived,
            n_rows=p.n_rows,
            n_targets=p.n_targets,
            n_out=p.n_out,
            reserved=0,
            free_point=i32(p.free_point),
            dop_type=i32(p.dop_type),
            dop_out=i32(p.dop_out),
            dop_pts=i32(p.dop_pts),
            dop_param=f64(p.dop_param),
            row_type=i32(p.row_type),
            row_pts=i32(p.row_pts),
            row_param=f64(p.row_param),
            tgt_point=i32(p.tgt_point),
            tgt_dir=f64(p.tgt_dir),
            out_point=i32(p.out_point),
            design_pos=f64(p.design_pos),
        )

    def byref(self):
        return C.byref(self.desc)
