#!/usr/bin/env python3
"""C2 cold sweep: plain back-to-back launches against one HIP graph holding K launches (what the launch gap costs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
dev = torch.device("cuda", 0)
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, dev)
t = torch.as_tensor(targets, device=dev)
out = torch.empty((16384, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((16384, 40), dtype=torch.uint8, device=dev)
launch = dp.plan(t, out=out, info_out=info, chain_len=-1, predictor=False)
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for _ in range(500): launch()
print(f"plain launches: {timed(launch, 2000) * 1e3:.2f} us per sweep")
for K in (1, 10, 100):
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        side_launch = dp.plan(t, out=out, info_out=info, chain_len=-1, predictor=False)  # a plan binds the stream it is made on
        side_launch(); side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K): side_launch()
    ms = timed(g.replay, max(2000 // K, 20))
    print(f"graph of {K:3d} launches: {ms / K * 1e3:.2f} us per sweep")
