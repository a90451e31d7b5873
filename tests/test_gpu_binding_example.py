"""
GPU: the raw ctypes binding of INTEGRATION.md section 2, as a maintainer of the reference would write it - no
``DeviceProgram``, only ``libokx.so`` and the structs of ``include/okx.h`` - for the plain and the evaluated solve.
"""

import ctypes as C

import numpy as np
import pytest
import torch

from conftest import gpu_available

pytestmark = pytest.mark.gpu


def test_raw_ctypes_binding_of_the_solve_and_the_evaluated_solve(golden):
    if not gpu_available():
        pytest.skip("no GPU")
    from open_kinematics_amd import _abi, _lib
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import METRIC_NAMES, CornerRoles, corner_roles
    from open_kinematics_amd.workloads import bump_sweep_problem, geometry_path

    class SolveOpts(C.Structure):                     # == okx_solve_opts, as printed in INTEGRATION.md
        _fields_ = [("max_iter", C.c_int32), ("chain", C.c_int32), ("steps_per_geometry", C.c_int64),
                    ("chain_len", C.c_int64), ("step_tol", C.c_double), ("grad_tol", C.c_double), ("ftol", C.c_double),
                    ("lambda0", C.c_double), ("residual_tolerance", C.c_double), ("kernel", C.c_int32),
                    ("confirm_full_pass", C.c_int32), ("predictor", C.c_int32), ("shared_first_step", C.c_int32),
                    ("output", C.c_int32), ("reserved", C.c_int32)]

    assert C.sizeof(SolveOpts) == C.sizeof(_abi.SolveOpts)
    lib = C.CDLL(_lib.LIB_PATH)
    lib.okx_last_error.restype = C.c_char_p
    lib.okx_program_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.okx_program_destroy.argtypes = [C.c_void_p]
    lib.okx_program_ready.argtypes = [C.c_void_p, C.c_int32]
    lib.okx_default_opts.argtypes = [C.POINTER(SolveOpts)]
    lib.okx_solve_batch.argtypes = [C.c_void_p, C.POINTER(SolveOpts), C.c_int64] + [C.c_void_p] * 6
    lib.okx_program_enable_evaluation.argtypes = [C.c_void_p, C.POINTER(CornerRoles)]
    lib.okx_solve_evaluated_batch.argtypes = [C.c_void_p, C.POINTER(SolveOpts), C.c_int64] + [C.c_void_p] * 8
    assert lib.okx_abi_version() == _abi.ABI_VERSION == 6

    program, targets_host = bump_sweep_problem(512)
    host = _abi.HostProgram(program)                  # fills okx_program_desc from the flattened program
    torch.cuda.set_device(0)
    handle = C.c_void_p()
    assert lib.okx_program_create(host.byref(), C.byref(handle)) == 0, lib.okx_last_error()
    try:
        assert lib.okx_program_ready(handle, 1) == 1
        opts = SolveOpts()
        lib.okx_default_opts(C.byref(opts))
        b, t_count = targets_host.shape
        targets = torch.as_tensor(targets_host, device="cuda")
        out = torch.empty((b, program.n_out, 3), dtype=torch.float64, device="cuda")
        info = torch.empty((b, 40), dtype=torch.uint8, device="cuda")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.okx_solve_batch(handle, C.byref(opts), b, targets.data_ptr(), None, None, out.data_ptr(), info.data_ptr(), stream)
        assert rc == 0, lib.okx_last_error()
        torch.cuda.synchronize()
        flags = info.cpu().numpy().view(_abi.INFO_DTYPE).reshape(-1)["flags"]
        assert np.all((flags & 7) == 1)
        # the evaluated solve: roles once per program, then metrics only (no records written)
        roles = corner_roles(load_geometry(geometry_path("geometry.yaml")), program)
        assert lib.okx_program_enable_evaluation(handle, C.byref(roles)) == 0, lib.okx_last_error()
        ev = torch.empty((b, 1 + t_count, 24), dtype=torch.float64, device="cuda")
        info2 = torch.empty_like(info)
        opts.output = 2                               # OKX_OUTPUT_NONE
        rc = lib.okx_solve_evaluated_batch(handle, C.byref(opts), b, targets.data_ptr(), None, None, None, info2.data_ptr(),
                                           None, ev.data_ptr(), stream)
        assert rc == 0, lib.okx_last_error()
        torch.cuda.synchronize()
        assert torch.equal(info, info2)
        bump = t_count - 1
        wc = list(program.out_point).index(int(program.tgt_point[bump]))
        travel = ev[:, 0, METRIC_NAMES.index("wheel_travel")]
        assert float((travel - (out[:, wc, 2] - float(program.design_pos[program.tgt_point[bump]][2]))).abs().max()) <= 1e-12
        assert float((ev[:, 1 + bump, _abi.EVAL_RATE_WHEEL_CENTER_Z] - 1.0).abs().max()) <= 1e-9   # d hub z / d bump target
        bump_steer = ev[:, 1 + bump, METRIC_NAMES.index("roadwheel_angle")] / ev[:, 1 + bump, _abi.EVAL_RATE_WHEEL_CENTER_Z]
        assert bool(torch.isfinite(bump_steer).all())
    finally:
        lib.okx_program_destroy(handle)
