"""
The runtime-specialised "quad" kernel, CPU side: source generation from a constraint program,
hiprtc compilation into the on-disk cache (no device needed) and the fall-back decisions.
"""

import ctypes as C
import os

import pytest

from conftest import STEERED, UNSTEERED
from open_kinematics_amd import _abi, _lib


def _source(program) -> str:
    lib = _lib.load()
    host = _abi.HostProgram(program)
    size = lib.okx_quad_source(host.byref(), None, 0)
    if size < 0:
        raise ValueError(_lib.last_error())
    buf = C.create_string_buffer(size)
    assert lib.okx_quad_source(host.byref(), buf, size) == size
    return buf.value.decode()


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "u_dw_corner", "u_macpherson", "rows_all_classes"])
@pytest.mark.parametrize("mode", ["pinned", "softnorm"])
def test_source_is_generated_for_corner_topologies(golden, name, mode):
    _, program = golden(name)
    p = program.with_line_mode(mode)
    src = _source(p)
    for kernel in ("okx_quad_solve_u", "okx_quad_solve_g", "okx_quad_eval"):
        assert f"void __launch_bounds__(64, 1) {kernel}(" in src
    assert "void __launch_bounds__(64) okx_quad_expand(" in src  # positions from free coordinates
    if p.n_targets:
        assert "okx_quad_tangent_u(" in src and "a.predictor" in src  # tangents; chain-head model
    # one residual per row, one J^T J accumulator per diagonal block entry
    for i in range(p.n_residuals):
        assert f"const double r{i} =" in src
    for f in range(p.n_free):
        for k in range(3):
            assert f"double A{f}_{f}_{k}" in src
    # the per-problem decisions rely on bit-identical quad reductions
    assert "#pragma clang fp contract(off)" in src
    # structure only: no geometry value is baked into the text
    assert "471.69" not in src and "559.01" not in src and "410.0" not in src


def test_same_structure_gives_the_same_kernel_different_structure_does_not(golden):
    _, a = golden("c1_dw_corner")
    _, b = golden("c2_dw_subset")  # same topology, same geometry file, other targets
    _, u = golden("u_dw_corner")   # toe link instead of the rack: other rows
    assert _source(a.with_line_mode("pinned")) == _source(b.with_line_mode("pinned"))
    assert _source(a.with_line_mode("pinned")) != _source(u.with_line_mode("pinned"))
    assert _source(a.with_line_mode("pinned")) != _source(a.with_line_mode("softnorm"))


def test_large_programs_without_pair_structure_keep_the_generic_kernels(golden):
    _, program = golden("u_axle")  # 20 free points, the two corners are not joined by any row
    lib = _lib.load()
    host = _abi.HostProgram(program.with_line_mode("pinned"))
    assert lib.okx_quad_source(host.byref(), None, 0) == -2  # OKX_ERR_LIMIT
    assert "free points" in _lib.last_error() and "pair of identical halves" in _lib.last_error()
    assert lib.okx_precompile(host.byref()) == -2


def test_axle_is_generated_in_pair_mode(golden):
    """Two identical corners joined by the rack row: the half program is generated, one quad per half."""
    _, program = golden("c3_axle_grid")
    src = _source(program.with_line_mode("pinned"))
    assert "quad = lane >> 3, q1 = (lane >> 2) & 1" in src          # 8 problems per wavefront
    assert "__builtin_amdgcn_ds_swizzle" in src and "sm_k" in src   # cross-quad exchange, 2 x 2 Woodbury for the rack row
    for f in range(10):
        assert f"double A{f}_{f}_0" in src                           # ten free points per half
    assert "double A10_10_0" not in src
    assert "okx_quad_eval" not in src            # the parity hook: the interpreter serves it
    assert "okx_quad_expand(" in src and "fin[" in src  # round 5: the expand is generated (input rows staged through LDS)
    assert "okx_quad_tangent_u(" in src and "sm_det" in src          # tangents are generated (regularised halves)
    assert "xsl[" in src and "a.predictor[" not in src               # chain state in LDS; no chain-head model in pair mode


def test_pair_mode_constants_live_in_registers_unless_lds_homes_are_asked_for(golden, monkeypatch):
    """Chain constants and fixed points: registers by default (quad_build falls back to LDS homes when that variant
    spills), LDS with the experiment switch."""
    _, program = golden("c3_axle_grid")
    pinned = program.with_line_mode("pinned")
    src = _source(pinned)
    assert "psl[" not in src and " = hsl[" not in src and "const double hcL = gq[" in src
    monkeypatch.setenv("OKX_DEV", "pair_lds_homes")
    src = _source(pinned)
    assert "psl[" in src and " = hsl[" in src and "#define hcL hsl[" in src


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "c3_axle_grid"])
def test_solve_kernels_of_the_baseline_programs_do_not_spill(golden, name):
    """Private segment size of okx_quad_solve_u/_g from the code object's metadata (in-tree cache after build())."""
    _, program = golden(name)
    lib = _lib.load()
    host = _abi.HostProgram(program.with_line_mode("pinned"))
    scratch = C.c_int32(-7)
    assert lib.okx_debug_kernel_scratch(host.byref(), C.byref(scratch)) == 0, _lib.last_error()
    # the corner kernels: nothing; the axle's pair-mode kernel keeps its register-resident layout up to 256 B (prologue
    # temporaries around the first-step table, okx_jit.cpp quad_build) - the LDS-homes layout would cost a wavefront per CU
    assert scratch.value == 0 if name != "c3_axle_grid" else 0 <= scratch.value <= 256


def test_precompile_fills_the_cache_without_a_device(golden, tmp_path, monkeypatch):
    _, program = golden("c1_dw_corner")
    monkeypatch.setenv("OKX_KERNEL_CACHE", str(tmp_path))
    lib = _lib.load()
    host = _abi.HostProgram(program.with_line_mode("pinned"))
    assert lib.okx_precompile(host.byref()) == 0, _lib.last_error()
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".okxc"))
    # the quad kernel's code object and the lane kernel's (every emission variant the build had to try, okx_jit.cpp lane_build)
    assert len(files) >= 2
    blobs = [(tmp_path / f).read_bytes() for f in files]
    assert all(b[:6] == b"OKXCK1" and b[24:28] == b"\x7fELF" for b in blobs)
    assert sum(b"okx_quad_solve_u" in b for b in blobs) == 1 and sum(b"okx_lane_solve_u" in b for b in blobs) == len(files) - 1
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".lanevar")]) == 1  # the variant that was kept
    stamps = [os.path.getmtime(tmp_path / f) for f in files]
    assert lib.okx_precompile(host.byref()) == 0  # second call is a cache hit
    assert [os.path.getmtime(tmp_path / f) for f in files] == stamps


def test_header_documents_the_kernel_choice():
    from conftest import REPO

    text = open(os.path.join(REPO, "include", "okx.h"), encoding="utf-8").read()
    for name in ("okx_program_kernel", "okx_quad_source", "okx_precompile"):
        assert name in text


def test_first_step_table_kernels_are_generated_where_they_pay(golden, monkeypatch):
    """okx_quad_head_u/_g: single-mode programs with targets carry them and take the unit's head step
    in the prologue; pair-mode kernels only on request; OKX_DEV=quad_no_head removes them."""
    _, dw = golden("c1_dw_corner")
    src = _source(dw.with_line_mode("pinned"))
    assert "okx_quad_head_u(QHeadArgs a)" in src and "okx_quad_head_g(QHeadArgs a)" in src
    assert "__shared__ double hxl[" in src and "if (head_ready && b == first_b)" in src
    # table stride = 4 n_free (T + 1) + 2 (T + 1)^2 + 8 doubles + the second-order vectors, 4 n_free per target pair
    k = dw.n_targets + 1
    stride = 4 * dw.n_free * k + 2 * k * k + 8 + 4 * dw.n_free * dw.n_targets * (dw.n_targets + 1) // 2
    assert "hS0_" in src and "hR0_" in src and "hw2" in src  # second-order terms: tabulated by the head kernel, used by the solve
    assert f"double* ho = a.head + geom * {stride};" in src
    assert "v_div" not in src and " / pred" not in src  # control code divides through refined reciprocals
    _, axle = golden("c3_axle_grid")
    pair = _source(axle.with_line_mode("pinned"))
    # pair mode: LDS state; the table with both halves' second-order vectors, taken in the unit prologue like the corner's
    assert "okx_quad_head_u(QHeadArgs a)" in pair and "hS0_" in pair and "lms[" in pair and "lean_atan2_pos<true>" in pair
    assert "double* hso = hs + (q1 ?" in pair and "if (head_ready && b == first_b)" in pair
    k = axle.n_targets + 1
    side_free = axle.n_free // 2
    stride = 2 * 4 * side_free * k + 2 * k * k + 8 + 2 * 4 * side_free * axle.n_targets * (axle.n_targets + 1) // 2
    assert f"double* hs = a.head + geom * {stride};" in pair
    monkeypatch.setenv("OKX_DEV", "pair_first_order_head")
    assert "hS0_" not in _source(axle.with_line_mode("pinned"))
    monkeypatch.setenv("OKX_DEV", "pair_no_head")
    assert "okx_quad_head_u(QHeadArgs" not in _source(axle.with_line_mode("pinned"))
    monkeypatch.setenv("OKX_DEV", "quad_no_head")
    assert "okx_quad_head_u(QHeadArgs" not in _source(dw.with_line_mode("pinned"))


def test_generated_sources_are_the_same_in_every_process(tmp_path):
    """The kernel cache is keyed by the source text: a stray printf conversion in a generated comment once made the quad
    source differ from process to process, and every program creation compiled its kernels afresh (2.7 s)."""
    import subprocess
    import sys

    from conftest import REPO

    code = (
        "import sys, hashlib, ctypes as C\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "from open_kinematics_amd import _lib\n"
        "from open_kinematics_amd._abi import HostProgram\n"
        "from open_kinematics_amd.workloads import bump_sweep_problem, macpherson_grid_problem, axle_grid_problem\n"
        "lib = _lib.load()\n"
        "for make in (bump_sweep_problem, lambda n: macpherson_grid_problem(n, n), lambda n: axle_grid_problem(n, n)):\n"
        "    hp = HostProgram(make(4)[0])\n"
        "    for fn in (lib.okx_quad_source, lib.okx_lane_source):\n"
        "        n = fn(hp.byref(), None, 0)\n"
        "        if n < 0:\n"
        "            print('none'); continue\n"
        "        buf = C.create_string_buffer(n); fn(hp.byref(), buf, n)\n"
        "        print(hashlib.sha256(buf.value).hexdigest())\n"
    )
    runs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300) for _ in range(2)]
    assert all(r.returncode == 0 for r in runs), runs[0].stderr[-1000:]
    assert runs[0].stdout == runs[1].stdout and len(runs[0].stdout.split()) == 6
