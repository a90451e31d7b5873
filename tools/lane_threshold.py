#!/usr/bin/env python3
"""Quad kernel against lane kernel for batch sizes between one quad round (16384 problems) and one lane round (65536):
where auto selection should switch.   python3 tools/lane_threshold.py [dw|mac]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd import workloads as W
from open_kinematics_amd.batch import DeviceProgram
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "dw"
for n in (8192, 12288, 16384, 17408, 20480, 24576, 32768, 49152, 65536, 98304):
    if which == "dw":
        program, t = W.bump_sweep_problem(n)
    else:
        import math
        e = int(math.isqrt(n))
        program, t = W.macpherson_grid_problem(e, n // e)
    dp = DeviceProgram(program, dev)
    tg = torch.as_tensor(t, device=dev)
    m = tg.shape[0]
    out = torch.empty((m, program.n_out, 3), dtype=torch.float64, device=dev)
    info = torch.empty((m, 40), dtype=torch.uint8, device=dev)
    row = []
    for cl in (1, -1):
        for kern in ("quad", "lane"):
            launch = dp.plan(tg, out=out, info_out=info, chain_len=cl, predictor=False, kernel=kern)
            for _ in range(50): launch()
            wall, ms = bench.time_launches(launch, 200, 10, dev)
            row.append(ms * 1e3)
    print(f"{which} B={m:6d}: cold quad {row[0]:7.2f} us lane {row[1]:7.2f} us | chained quad {row[2]:7.2f} us lane {row[3]:7.2f} us")
    dp.close()
