#!/usr/bin/env python3
"""Host buffers in, host buffers out with the solve kernel storing its compact output (free coordinates + info) straight
into pinned, device-mapped host memory: no D2H copy, the stores cross PCIe as the kernel issues them.  Compared with the
copy-based pipeline of bench.measure_e2e_compact.   python3 tools/e2e_zero_copy.py [streams] [sweeps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

dev = torch.device("cuda", 0)
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, dev)
n = targets.shape[0]
kw = dict(chain_len=-1, predictor=False)
h_t = torch.as_tensor(targets).pin_memory()
for targets_mapped in (False, True):
    slots = []
    for _ in range(n_slots):
        stream = torch.cuda.Stream(dev)
        h_free = torch.empty((n, program.n_free, 3), dtype=torch.float64).pin_memory()
        h_info = torch.empty((n, 40), dtype=torch.uint8).pin_memory()
        with torch.cuda.stream(stream):
            d_t = torch.empty_like(h_t, device=dev)
            launch = dp.plan(h_t if targets_mapped else d_t, out=h_free, info_out=h_info, output="free", zero_copy=True, **kw)
        slots.append(dict(stream=stream, d_t=d_t, launch=launch, h_free=h_free, h_info=h_info, done=torch.cuda.Event()))

    def issue(slot):
        with torch.cuda.stream(slot["stream"]):
            if not targets_mapped:
                slot["d_t"].copy_(h_t, non_blocking=True)
            slot["launch"]()
            slot["done"].record()

    for k in range(2 * n_slots):
        issue(slots[k % n_slots])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        slot = slots[k % n_slots]
        slot["done"].synchronize()
        issue(slot)
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / steps
    info = slots[0]["h_info"].numpy().view(bench.INFO_FIELDS).reshape(-1)
    ref = dp.solve(torch.as_tensor(targets, device=dev), output="free", **kw)
    d = float(np.abs(slots[0]["h_free"].numpy() - ref.free.cpu().numpy()).max())
    print(json.dumps({"zero_copy_out": True, "targets_read_over_pcie": targets_mapped, "streams": n_slots, "value": n / wall,
                      "us_per_sweep": wall * 1e6, "all_converged": bool(np.all((info["flags"] & 7) == 1)), "max_diff_vs_device_buffers": d}))
print(json.dumps(bench.measure_e2e_compact(dp, targets, dev, steps, kw)))
