#!/usr/bin/env python3
"""Driver-style timing of K = 20 C2 steps between two synchronisations: K stream launches against ONE launch of a HIP graph
holding the same K kernel nodes (captured ahead of the timed region).  python3 tools/graph_steps.py [K]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, dev)
t = torch.as_tensor(targets, device=dev)
out = torch.empty((16384, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((16384, 40), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    launch = dp.plan(t, out=out, info_out=info, chain_len=1, predictor=False)
    for _ in range(50):
        launch()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(K):
            launch()
    side.synchronize()


def timed(fn, reps=30):
    best = []
    for _ in range(reps):
        t_end = time.perf_counter() + 0.04          # the GPU out of idle, as bench.py's preheat does
        with torch.cuda.stream(side):
            while time.perf_counter() < t_end:
                launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best.append(time.perf_counter() - t0)
    return np.median(best) * 1e6, np.min(best) * 1e6


def plain():
    with torch.cuda.stream(side):
        for _ in range(K):
            launch()


for label, fn in (("stream launches", plain), ("one graph launch", graph.replay), ("stream launches", plain), ("one graph launch", graph.replay)):
    med, lo = timed(fn)
    print(f"{label:18s}: {med / K:7.2f} us per step (median of 30; best {lo / K:.2f}) -> {16384 / (med / K) * 1e6:.3e} solves/s")
