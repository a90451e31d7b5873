"""
The lane kernel (one lane per problem, okx_lanegen.cpp), CPU side: source generation from a constraint program and the
decisions about which programs get one.  (Compilation into the cache is part of __graft_entry__.build().)
"""

import ctypes as C

import pytest

from open_kinematics_amd import _abi, _lib


def _lane_source(program) -> str:
    lib = _lib.load()
    host = _abi.HostProgram(program)
    size = lib.okx_lane_source(host.byref(), None, 0)
    if size < 0:
        raise ValueError(_lib.last_error())
    buf = C.create_string_buffer(size)
    assert lib.okx_lane_source(host.byref(), buf, size) == size
    return buf.value.decode()


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "u_dw_corner", "u_macpherson", "rows_all_classes"])
@pytest.mark.parametrize("mode", ["pinned", "softnorm"])
def test_lane_source_is_generated_for_corner_topologies(golden, name, mode):
    _, program = golden(name)
    p = program.with_line_mode(mode)
    src = _lane_source(p)
    for kernel in ("okx_lane_solve_u", "okx_lane_solve_g", "okx_lane_chain_u", "okx_lane_chain_g", "okx_lane_eval"):
        assert f"void __launch_bounds__(64, 1) {kernel}(" in src
    n = p.n_vars
    # one residual per row; the diagonal of J^T J is accumulated at the rows, every other entry is assembled when its
    # column is eliminated (left-looking LDL^T); one pivot per unknown
    for i in range(p.n_residuals):
        assert f"const double r{i} =" in src
    for i in range(n):
        assert f"A{i}_{i} = 0.0" in src and f"const double dinv{i} = pivot_rcp(C{i}_{i});" in src
    assert "__builtin_amdgcn_mov_dpp" not in src and "ds_swizzle" not in src  # no cross-lane operand anywhere
    # a table entry's name is a macro over GL(offset).  Own geometry and chains: wave-uniform reads through the scalar cache
    # (constant-address-space copies of the table pointers, an opaque scalar offset); independent solves on per-geometry
    # tables: a body of its own that stages them into LDS once per geometry, the loads in one batch
    assert "#define hs0_0 GL(" in src and "const okx_cptr gpc = (okx_cptr)gp" in src and '"+s"(kzs)' in src
    assert "void okx_lane_body_coldg(const QArgs& a)" in src and "#define GL(o) gl[(o) + kz]" in src
    assert "okx_lane_solve_g(QArgs a) { okx_lane_body_coldg<true, true>(a); }" in src
    assert "if (span_idx != staged_span) {" in src and "gl[0 + lane] = sv0;" in src
    # structure only: no geometry value is baked into the text
    assert "471.69" not in src and "559.01" not in src and "410.0" not in src


def test_lane_kernel_shares_the_quad_kernels_elimination_order(golden):
    """The first-step tables come from okx_quad_head_*: both generators must number the free blocks alike."""
    import re

    _, program = golden("c1_dw_corner")
    p = program.with_line_mode("pinned")
    lane = _lane_source(p)
    lib = _lib.load()
    host = _abi.HostProgram(p)
    size = lib.okx_quad_source(host.byref(), None, 0)
    buf = C.create_string_buffer(size)
    lib.okx_quad_source(host.byref(), buf, size)
    quad = buf.value.decode()
    # quad: `double x{F} = p{point}` per block; lane: `x{3F} = p{point}_0`
    quad_order = [int(m) for m in re.findall(r"double x\d+ = p(\d+), xp\d+", quad)]
    lane_order = [int(m) for m in re.findall(r"\bx\d+ = p(\d+)_0; dx\d+ = 0\.0;", lane)]
    assert quad_order and quad_order == lane_order[: len(quad_order)]


def test_programs_beyond_six_free_points_have_no_lane_kernel(golden):
    _, program = golden("c3_axle_grid")  # 20 free points (pair mode in the quad kernel)
    lib = _lib.load()
    host = _abi.HostProgram(program.with_line_mode("pinned"))
    assert lib.okx_lane_source(host.byref(), None, 0) == -2  # OKX_ERR_LIMIT
    assert "free points" in _lib.last_error()
    assert lib.okx_precompile(host.byref()) == 0  # the quad kernel alone is precompiled then


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "c5_ensemble"])
def test_independent_solve_bodies_of_the_baseline_programs_do_not_spill(golden, name):
    """The lane kernel sits at the edge of the 512-register file: lane_build compiles its emission variants until one keeps
    the independent-solve bodies out of scratch (in-tree cache after build()).  The chain bodies of the 18-unknown double
    wishbone still spill (auto selection keeps the quad kernel's chains for it)."""
    _, program = golden(name)
    lib = _lib.load()
    host = _abi.HostProgram(program.with_line_mode("pinned"))
    out = (C.c_int32 * 3)()
    assert lib.okx_debug_lane_scratch(host.byref(), out) == 0, _lib.last_error()
    # (the worst of the four kernels as they are launched - a kernel that spills less in another variant's module is taken
    #  from there.  MacPherson: nothing spills.  The double wishbone: the full-record kernels keep everything in registers,
    #  the compact per-geometry one keeps 20 B of prologue values in scratch in its best variant - measured harmless,
    #  where 100 B inside the passes cost a quarter of the rate: profiles/r04/EXPERIMENTS.md section 6)
    assert out[0] <= (0 if name == "c4_macpherson_grid" else 32), f"independent-solve bodies spill {out[0]} B (variant {out[2]})"
    if name == "c4_macpherson_grid":
        assert out[1] == 0
    # the choice (kept variant + the kernels taken from other variants) is remembered beside the code objects: asking again
    # reads it back instead of searching, and must come to the same kernels
    again = (C.c_int32 * 3)()
    assert lib.okx_debug_lane_scratch(host.byref(), again) == 0, _lib.last_error()
    assert list(again) == list(out)


def test_small_programs_keep_the_independent_solve_state_in_registers(golden):
    """Programs of up to 15 variables (the MacPherson corner) have registers to spare: the independent-solve body of emission
    variants 0 - 11 keeps the accepted point, the step in hand and the whole factor in registers (an LDS round trip costs a
    lone wavefront more than the two moves of a parked register: profiles/r04/EXPERIMENTS.md section 11); the chain body and
    every larger program keep the LDS layout."""
    _, mac = golden("c4_macpherson_grid")
    src = _lane_source(mac.with_line_mode("pinned"))
    cold = src[src.index("void okx_lane_body_cold(const QArgs& a)"):src.index("void okx_lane_body_coldg(const QArgs& a)")]
    chain = src[src.index("void okx_lane_body_chain(const QArgs& a)"):]
    assert mac.n_vars == 15
    assert "    double x0, dx0;" in cold and "double& x0 = lds[" not in cold
    coldg = src[src.index("void okx_lane_body_coldg(const QArgs& a)"):src.index("void okx_lane_body_chain(const QArgs& a)")]
    assert "    double x0, dx0;" in coldg and "double& x0 = lds[" not in coldg  # (per-geometry tables: the same layout)
    assert "double& x0 = lds[" in chain and "double& xp0 = lds[" in chain
    _, dw = golden("c1_dw_corner")
    src = _lane_source(dw.with_line_mode("pinned"))
    cold = src[src.index("void okx_lane_body_cold(const QArgs& a)"):src.index("void okx_lane_body_coldg(const QArgs& a)")]
    assert dw.n_vars == 18 and "double& x0 = lds[" in cold and "    double x0, dx0;" not in cold
