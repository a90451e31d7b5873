#!/usr/bin/env python3
"""Where the host-side microseconds of a 20-step timed region go (bench.py --steps 20: ms_per_step exceeds kernel_ms by 0.8 - 3.4 us
per step depending on the box): event record, graph launch, end-event record, until a poll sees the end event, synchronize."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
p, t = bump_sweep_problem(16384)
dp = DeviceProgram(p, dev)
tt = torch.as_tensor(t, device=dev)
out = torch.empty((16384, p.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((16384, 40), dtype=torch.uint8, device=dev)
launch = dp.plan(tt, out=out, info_out=info, chain_len=1, predictor=False)
for _ in range(2000):
    launch()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=torch.cuda.current_stream(dev)):
    for _ in range(K):
        launch()
g.replay()
torch.cuda.synchronize()
rows = []
for mode in ("graph", "stream"):
    for rep in range(12):
        for _ in range(200):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        t1 = time.perf_counter()
        if mode == "graph":
            g.replay()
        else:
            for _ in range(K):
                launch()
        t2 = time.perf_counter()
        e1.record()
        t3 = time.perf_counter()
        while not e1.query():
            pass
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        rows.append((mode, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6, (t5 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
for mode in ("graph", "stream"):
    a = np.array([r[1:] for r in rows if r[0] == mode][2:])
    m = np.median(a, axis=0)
    print(f"{mode:6s} K={K}: start-event record {m[0]:.1f} us, submit {m[1]:.1f}, end-event record {m[2]:.1f}, until the poll sees the end {m[3]:.1f}, "
          f"synchronize {m[4]:.1f}; wall {m[5]:.1f} us = {m[5] / K:.2f} per step; GPU start -> end event {m[6]:.1f} us = {m[6] / K:.2f} per step")
