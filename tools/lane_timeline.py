#!/usr/bin/env python3
"""Where a wave unit of the lane kernel's independent-solve body spends its cycles (a build generated with
OKX_DEV=lane_timeline stamps the shader clock per wave unit: unit start, tables staged, first step in hand, passes done, final
state, info stored, records stored; and keeps the section times of the unit's last full pass).
   OKX_DEV=lane_timeline OKX_KERNEL_CACHE=build/kc_tl python3 tools/lane_timeline.py [c4 | c5 | c2x16]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd import _lib
_lib.LIB_PATH = os.environ.get("LANE_TL_LIB", _lib.LIB_PATH)   # (A/B against another build of the library)
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem, ensemble_problem, macpherson_grid_problem

what = sys.argv[1] if len(sys.argv) > 1 else "c4"
nest = what.endswith("nest")   # c5nest / c2x16nest: the nested start mode (chain_len = -1; one trace row per unit-step)
what = what[:-4] if nest else what
chain_len = -1 if nest else 1
dev = torch.device("cuda:0")
kw = {}
if what == "c5":
    program, table, rel = ensemble_problem(4096, 256)
    dp = DeviceProgram(program, dev)
    gpos, gparam = dp.rebind(torch.as_tensor(table, device=dev))
    t = dp.ensemble_targets(gpos, rel)
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=rel.shape[0])
elif what == "c4":
    program, targets = macpherson_grid_problem(512, 512)
    dp = DeviceProgram(program, dev)
    t = torch.as_tensor(targets, device=dev)
else:
    program, targets = bump_sweep_problem(16 * 16384)
    dp = DeviceProgram(program, dev)
    t = torch.as_tensor(targets, device=dev)
n = t.shape[0]
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)
spg = kw.get("steps_per_geometry", 0)
units = (n // spg) * ((spg + 63) // 64) if spg else (n + 63) // 64
if nest:
    units = 4 * ((n // spg) * ((spg // 4 + 63) // 64) if spg else (n // 4 + 63) // 64)
print(f"{what}: {n} problems, {units} wave units, lane kernel from {dp.lane_threshold} problems, bodies {dp.lane_bodies}")

launch = dp.plan(t, out=out, info_out=info, chain_len=chain_len, predictor=False, kernel="lane", **kw)
for _ in range(20):
    launch()
wall, ms = bench.time_launches(launch, 50, 5, dev)
print(f"{1e3 * ms:.1f} us per launch ({n / (ms * 1e-3):.3e} solves/s)")
free = torch.empty((n, program.n_vars // 3, 3), dtype=torch.float64, device=dev)
launch_free = dp.plan(t, out=free, info_out=info, chain_len=chain_len, predictor=False, kernel="lane", output="free", **kw)
for _ in range(20):
    launch_free()
wall, ms_free = bench.time_launches(launch_free, 50, 5, dev)
print(f"{1e3 * ms_free:.1f} us per launch with output = free (the compact kernel)")
if "lane_timeline" not in os.environ.get("OKX_DEV", ""):
    sys.exit(0)
tr = torch.zeros((units, 32), dtype=torch.float64, device=dev)
dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), -1)
launch = dp.plan(t, out=out, info_out=info, chain_len=chain_len, predictor=False, kernel="lane", **kw)
for _ in range(5):
    launch()
torch.cuda.synchronize()
tr.zero_()
launch()
torch.cuda.synchronize()
a = tr.cpu().numpy()
dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
# (the XCDs' shader clocks have different bases: only differences inside one wave unit are used)
stamps = [(0, "unit start"), (1, "tables staged in LDS"), (16, "state set up (x, dx in LDS)"), (2, "first step in hand"), (3, "LM passes done"),
          (13, "final state"), (14, "info stored"), (17, "records transposed in LDS"), (15, "records stored")]
print("cycles from the previous stamp: median / min / max over the wave units")
for (k0, _), (k1, name) in zip(stamps[:-1], stamps[1:]):
    d = a[:, k1] - a[:, k0]
    print(f"  {name:30s} {np.median(d):9.0f} {d.min():9.0f} {d.max():9.0f}")
life = a[:, 15] - a[:, 0]
print(f"wave unit, start -> records stored: median {np.median(life):.0f}, min {life.min():.0f}, max {life.max():.0f}")
full, light = a[:, 4], a[:, 5]
print("full passes per wave unit:", dict(zip(*[x.tolist() for x in np.unique(full, return_counts=True)])),
      " confirming passes:", dict(zip(*[x.tolist() for x in np.unique(light, return_counts=True)])))
lm = a[:, 3] - a[:, 2]
print(f"cycles per pass (LM loop / passes of either kind): median {np.median(lm / np.maximum(full + light, 1)):.0f}")
print("sections of a unit's last full pass (median cycles):")
for k, name in ((6, "pass top -> trial point"), (7, "rows: residuals, gradients, J^T r, J^T J"), (8, "LM decision"), (9, "LDL^T factorisation"),
                (10, "substitutions"), (11, "step norms, next-pass logic")):
    print(f"  {name:44s} {np.median(a[:, k]):8.0f}")
print(f"  {'sum':44s} {np.median(a[:, 6:12].sum(axis=1)):8.0f}")
# consecutive units of one wavefront: end of one unit -> start of the next (grid-stride loop: unit wu + gridDim)
