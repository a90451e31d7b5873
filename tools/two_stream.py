#!/usr/bin/env python3
"""C2 cold sweeps launched back to back on one stream vs alternating over two streams / output buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
p, t = bump_sweep_problem(16384)
dev = torch.device("cuda:0")
dp = DeviceProgram(p, dev)
tg = torch.as_tensor(t, device=dev)
K = 400
for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    plans = []
    for s in streams:
        with torch.cuda.stream(s):
            out = torch.empty((16384, p.n_out, 3), dtype=torch.float64, device=dev)
            info = torch.empty((16384, 40), dtype=torch.uint8, device=dev)
            plans.append(dp.plan(tg, out=out, info_out=info, chain_len=-1, predictor=False))
    for k in range(20): plans[k % n_streams]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): plans[k % n_streams]()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{n_streams} stream(s): {dt*1e6:.2f} us per sweep, {16384/dt:.4g} solves/s")
