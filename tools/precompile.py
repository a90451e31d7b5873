#!/usr/bin/env python3
"""Compile the generated kernels of BASELINE programs into the kernel cache (no GPU needed; honours OKX_KERNEL_CACHE and the
generator's developer switches):  tools/precompile.py dw|mac|axle ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd import _lib
from open_kinematics_amd._abi import HostProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem

lib = _lib.load()
for w in sys.argv[1:] or ["dw"]:
    program = {"dw": lambda: bump_sweep_problem(5)[0], "mac": lambda: macpherson_grid_problem(4, 4)[0],
               "axle": lambda: axle_grid_problem(4, 4)[0]}[w]()
    rc = lib.okx_precompile(HostProgram(program).byref())
    print("precompile", w, rc, _lib.last_error() if rc else "")
