"""C-ABI checks that need no GPU: header <-> Python constants, exported symbols, plan building."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO, STEERED, UNSTEERED
from open_kinematics_amd import _abi, _lib, program as prog_mod

HEADER = os.path.join(REPO, "include", "okx.h")


def _header_text():
    with open(HEADER, "r", encoding="utf-8") as fh:
        return fh.read()


def test_enum_codes_match_header():
    text = _header_text()
    for name, value in re.findall(r"OKX_(ROW_[A-Z_]+|DOP_[A-Z_]+) = (\d+)", text):
        if name.endswith("TYPE_COUNT"):
            continue
        assert getattr(prog_mod, name) == int(value), name
    for macro, attr in [("OKX_MAX_VARS", "MAX_VARS"), ("OKX_MAX_ROWS", "MAX_ROWS"),
                        ("OKX_MAX_POINTS", "MAX_POINTS"), ("OKX_MAX_TARGETS", "MAX_TARGETS"),
                        ("OKX_ROW_PARAMS", "ROW_PARAMS"), ("OKX_ROW_POINTS", "ROW_POINTS")]:
        value = int(re.search(rf"#define {macro} (\d+)", text).group(1))
        assert getattr(prog_mod, attr) == value
    assert int(re.search(r"#define OKX_ABI_VERSION (\d+)", text).group(1)) == _abi.ABI_VERSION


def test_library_exports_every_declared_entry_point():
    lib = _lib.load()
    declared = set(re.findall(r"\b(okx_[a-z_]+)\s*\(", _header_text()))
    declared -= {"okx_program_desc", "okx_solve_opts", "okx_info", "okx_status", "okx_row_type", "okx_dop_type"}
    assert declared, "header parse failed"
    for name in declared:
        assert hasattr(lib, name), f"libokx.so does not export {name}"
    assert lib.okx_abi_version() == _abi.ABI_VERSION
    assert set(_lib.EXPORTS) == declared, "open_kinematics_amd._lib.EXPORTS and include/okx.h disagree"
    opts = _abi.SolveOpts()
    lib.okx_default_opts(C.byref(opts))
    assert opts.max_iter == 100 and opts.step_tol == 1e-11 and opts.residual_tolerance == 1e-3
    assert C.sizeof(_abi.Info) == 40


def test_library_exports_nothing_undeclared():
    """Every okx_* symbol of libokx.so is declared in include/okx.h or include/okx_debug.h."""
    import subprocess

    with open(os.path.join(REPO, "include", "okx_debug.h"), "r", encoding="utf-8") as fh:
        debug = set(re.findall(r"\b(okx_debug_[a-z_]+)\s*\(", fh.read()))
    assert debug == set(_lib.DEBUG_EXPORTS)
    declared = set(re.findall(r"\b(okx_[a-z_]+)\s*\(", _header_text())) | debug
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if line.split()[-1].startswith("okx_")}
    assert exported, "nm found no okx_* symbols"
    assert exported <= declared, f"undeclared exports: {sorted(exported - declared)}"
    lib = _lib.load()
    for name in debug:
        assert hasattr(lib, name), f"libokx.so does not export {name}"


@pytest.mark.parametrize("name", STEERED + UNSTEERED)
def test_plan_building_on_host(golden, name):
    _, program = golden(name)
    lib = _lib.load()
    for mode in ("softnorm", "pinned"):
        p = program.with_line_mode(mode)
        host = _abi.HostProgram(p)
        stats = (C.c_int32 * 8)()
        assert lib.okx_debug_plan_stats(host.byref(), stats) == 0
        n, m, pairs, contrib, active, js_stride, lda, lds = list(stats)
        assert (n, m) == (p.n_vars, p.n_residuals)
        assert pairs >= p.n_free and contrib >= m
        assert lda >= n and lda % 4 == 2 and js_stride % 2 == 1
        assert lds <= 160 * 1024


def test_invalid_programs_are_rejected_with_status_codes(golden):
    _, program = golden("u_dw_corner")
    lib = _lib.load()
    stats = (C.c_int32 * 8)()
    # underdetermined: OKX_ERR_UNDERDETERMINED and the reference's message (solver.py:116-121)
    few = program.with_targets([], np.zeros((0, 3)))
    few.row_type, few.row_pts, few.row_param, few.row_source = (
        few.row_type[:5], few.row_pts[:5], few.row_param[:5], few.row_source[:5])
    with pytest.raises(ValueError, match="System is underdetermined"):
        _abi.HostProgram(few)
    # bad ABI version
    good = _abi.HostProgram(program)
    good.desc.abi_version = 99
    assert lib.okx_debug_plan_stats(good.byref(), stats) == -1
    assert b"ABI version" in lib.okx_last_error()
    # out-of-range point index
    broken = program.with_targets(program.tgt_point, program.tgt_dir)
    broken.row_pts = program.row_pts.copy()
    broken.row_pts[0, 0] = 1000
    assert lib.okx_debug_plan_stats(_abi.HostProgram(broken).byref(), stats) == -1


def test_no_gpu_means_loud_failure():
    """The product path has no CPU fallback."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from conftest import load_golden
    from open_kinematics_amd.batch import DeviceProgram

    _, program = load_golden("u_dw_corner")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DeviceProgram(program)
    assert _lib.device_count() == 0
