"""
CPU: the camber-shim kernel source (okx_shim.hip) compiled for the host and run under ASan + UBSan on the
reference's setup states.  The functions are ``__host__ __device__``; this driver is test infrastructure only.
"""

import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import yaml

from conftest import REPO
from test_shims_oracle import CASES, load_shim_golden

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    exe = str(tmp_path_factory.mktemp("shim_host") / "shim_host")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-Wno-option-ignored", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-o", exe, "-x", "hip", os.path.join(REPO, "tests", "cpu_harness", "shim_host.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return exe


@pytest.mark.parametrize("case", CASES)
def test_kernel_source_on_the_host_matches_the_reference(driver, tmp_path, case):
    from open_kinematics_amd.input import build_suspension
    from open_kinematics_amd.shims import SHIM_INFO_DTYPE, shim_roles

    g = load_shim_golden(case)
    sus = build_suspension(yaml.safe_load(str(g["geometry_yaml"])))
    keys = list(sus.hardpoints)
    rows = [g["names"].index(k.name.lower()) for k in keys]
    n = len(g["setup"])
    table = np.ascontiguousarray(np.repeat(g["authored"][rows][None], n, axis=0))
    shim = np.ascontiguousarray(np.stack([sus.camber_shim.row(t) for t in g["setup"]]))
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as fh:
        fh.write(bytes(shim_roles(sus, keys)))
        fh.write(np.int64(n).tobytes() + np.int32(len(rows)).tobytes() + np.int32(0).tobytes())
        fh.write(table.tobytes() + shim.tobytes())
    env = {**os.environ, "ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "halt_on_error=1"}
    run = subprocess.run([driver, src, dst], capture_output=True, timeout=300, env=env)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    raw = open(dst, "rb").read()
    got = np.frombuffer(raw[: table.nbytes], dtype=np.float64).reshape(table.shape)
    info = np.frombuffer(raw[table.nbytes:], dtype=SHIM_INFO_DTYPE)
    assert np.max(np.abs(got - g["positions"][:, rows])) <= 1e-9
    assert np.all(info["converged"] == 1) and np.max(info["max_residual"]) <= 1e-10
    assert np.max(np.abs(info["rocker_angle_rad"] - g["rocker_angle"])) <= 1e-12
    assert C.sizeof(type(shim_roles(sus, keys))) == 116
