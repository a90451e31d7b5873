"""
CPU: the oracle's C restatement (oracle/okx_oracle.c) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer and run
over the committed goldens in a child interpreter (the sanitizer runtime has to be loaded first, hence LD_PRELOAD).
Every entry point is exercised — eval, positions, sweep (warm and cold), rebind — on every topology, and the results
must equal the regular build's bit for bit (same source, -ffp-contract=off in both).
"""

import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO

_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, {repo!r})
sys.path.insert(0, {repo!r} + "/tests")
from conftest import load_golden
from oracle.oracle import Oracle
out = {{}}
for name in {names!r}:
    arrays, program = load_golden(name)
    digest = []
    for mode in ("softnorm", "pinned"):
        orc = Oracle(program.with_line_mode(mode))
        r, j = orc.eval(arrays["eval_x"], arrays["eval_targets"])
        digest += [float(np.abs(r).sum()), float(np.abs(j).sum())]
        if "targets_abs" not in arrays:  # the synthetic all-classes program has no sweep
            continue
        t = arrays["targets_abs"][:24]
        warm = orc.sweep(t)
        cold = orc.sweep(t[:6], 1e-15, 1e-15, 1e-15, warm_start=False)
        digest += [float(warm.positions.sum()), float(cold.positions.sum()), int(warm.info["nfev"].sum()), int(warm.first_failed_step)]
        digest.append(float(orc.positions(warm.x[-1]).sum()))
        pos, rp = orc.rebind(orc.program.design_pos + 0.25)
        digest += [float(pos.sum()), float(rp.sum())]
    out[name] = digest
print(json.dumps(out))
"""

NAMES = ["c1_dw_corner", "c3_axle_grid", "c4_macpherson_grid", "u_dw_corner", "rows_all_classes"]


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    run = subprocess.run([sys.executable, "-c", _CHILD.format(repo=REPO, names=NAMES)], env=env, capture_output=True,
                         text=True, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    return json.loads(run.stdout.strip().splitlines()[-1])


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not found")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    lib = str(tmp_path / "libokx_oracle_san.so")
    subprocess.run([gcc, "-O1", "-g", "-fPIC", "-std=c11", "-Wall", "-Wextra", "-ffp-contract=off", "-fno-fast-math",
                    "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", lib,
                    os.path.join(REPO, "oracle", "okx_oracle.c"), "-lm"], check=True, capture_output=True, timeout=600)
    sanitized = _run({"OKX_ORACLE_LIB": lib, "LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0",
                      "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    regular = _run({})
    assert sanitized == regular
