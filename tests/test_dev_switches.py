"""
The developer switches of the kernel generators (one environment variable, ``OKX_DEV``; ``csrc/okx_quad.hpp``): every
switch that changes generated source still yields source that is deterministic (same text from call to call - the text is
the kernel-cache key), differs from the default build where it should, and passes the device compiler's front end
(``hipcc -fsyntax-only`` for gfx950: seconds, where a full hiprtc build of one module takes a minute).  CPU only.
"""

import ctypes as C
import os
import shutil
import subprocess

import pytest

from open_kinematics_amd import _abi, _lib

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _source(program, entry: str = "okx_quad_source") -> str:
    lib = _lib.load()
    host = _abi.HostProgram(program)
    fn = getattr(lib, entry)
    size = fn(host.byref(), None, 0)
    if size < 0:
        raise ValueError(_lib.last_error())
    buf = C.create_string_buffer(size)
    assert fn(host.byref(), buf, size) == size
    return buf.value.decode()


def _front_end_accepts(source: str, tmp_path, tag: str) -> None:
    path = tmp_path / f"{tag}.hip"
    path.write_text(source)
    proc = subprocess.run([HIPCC, "-fsyntax-only", "-x", "hip", "--offload-arch=gfx950", "--cuda-device-only", "-std=c++17",
                           "-include", "hip/hip_runtime.h", "-Wno-unused-command-line-argument", str(path)],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-3000:]


# switch, fixture, source entry point, text that must appear (+) or disappear (-) against the default build
GENERATOR_SWITCHES = [
    ("quad_mark", "c1_dw_corner", "okx_quad_source", "+s_nop 1"),
    ("quad_timeline", "c1_dw_corner", "okx_quad_source", "+__builtin_readcyclecounter"),
    ("quad_no_light", "c1_dw_corner", "okx_quad_source", "-want_light)) {"),
    ("quad_no_head", "c1_dw_corner", "okx_quad_source", "-okx_quad_head_u(QHeadArgs"),
    ("quad_no_fast", "c1_dw_corner", "okx_quad_source", "-bool hand_over ="),
    ("quad_two_waves", "c1_dw_corner", "okx_quad_source", "+__launch_bounds__(64, 2)"),
    ("pair_no_head", "c3_axle_grid", "okx_quad_source", "-okx_quad_head_u(QHeadArgs"),
    ("pair_first_order_head", "c3_axle_grid", "okx_quad_source", "-hS0_"),
    ("pair_lds_homes", "c3_axle_grid", "okx_quad_source", "+psl["),
    ("pair_cold_lds", "c3_axle_grid", "okx_quad_source", "-double Fc = 0.0, lambda = 0.0, nu = 2.0"),
    ("lane_mark", "c1_dw_corner", "okx_lane_source", "+s_nop 1"),
    ("lane_timeline", "c1_dw_corner", "okx_lane_source", "+__builtin_readcyclecounter"),
    ("lane_lds_tables", "c1_dw_corner", "okx_lane_source", "-okx_cptr gpc"),
    ("lane_nested", "c1_dw_corner", "okx_lane_source", "+okx_lane_nest_u"),
    ("lane_refine", "c1_dw_corner", "okx_lane_source", "+okx_lane_refw_u"),
]


@pytest.mark.parametrize("switch,fixture,entry,marker", GENERATOR_SWITCHES)
def test_generator_switches_give_deterministic_source_that_compiles(golden, monkeypatch, tmp_path, switch, fixture, entry, marker):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    _, program = golden(fixture)
    program = program.with_line_mode("pinned")
    monkeypatch.delenv("OKX_DEV", raising=False)
    default = _source(program, entry)
    monkeypatch.setenv("OKX_DEV", switch)
    first, second = _source(program, entry), _source(program, entry)
    assert first == second, "the same switch must give the same text every time"
    assert first != default
    needle = marker[1:]
    if marker[0] == "+":
        assert needle in first and needle not in default
    else:
        assert needle not in first and needle in default
    monkeypatch.setenv("OKX_DEV", f"unrelated,{switch}=1,other=3")  # list syntax: the switch is found among others
    assert _source(program, entry) == first
    monkeypatch.setenv("OKX_DEV", f"{switch}=0")                      # ... and `=0` leaves it off
    assert _source(program, entry) == default
    _front_end_accepts(first, tmp_path, switch)


def test_default_sources_pass_the_front_end(golden, tmp_path, monkeypatch):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    monkeypatch.delenv("OKX_DEV", raising=False)
    for fixture in ("c1_dw_corner", "c3_axle_grid"):
        _, program = golden(fixture)
        _front_end_accepts(_source(program.with_line_mode("pinned")), tmp_path, fixture)
    _, program = golden("c1_dw_corner")
    _front_end_accepts(_source(program.with_line_mode("pinned"), "okx_lane_source"), tmp_path, "lane")


def test_no_other_environment_switch_is_left_in_the_library():
    """The library reads OKX_DEV, OKX_KERNEL_CACHE and OKX_VERBOSE - nothing else (round 3 had ~50 getenv experiment switches)."""
    import re

    root = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "csrc")
    names = set()
    for name in os.listdir(root):
        if name.endswith((".cpp", ".hip", ".hpp")):
            with open(os.path.join(root, name), encoding="utf-8") as fh:
                names |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', fh.read()))
    assert names == {"OKX_DEV", "OKX_KERNEL_CACHE", "OKX_VERBOSE"}
