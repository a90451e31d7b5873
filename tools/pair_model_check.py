#!/usr/bin/env python3
"""C3 (rocker axle, pair mode) with and without the chain-head model (needs OKX_PAIR_MODEL=1 to be generated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
p, t = W.axle_grid_problem(256, 256)
dp = DeviceProgram(p, "cuda:0")
tg = torch.as_tensor(t, device="cuda:0")
out = torch.empty((tg.shape[0], p.n_out, 3), dtype=torch.float64, device="cuda:0")
info = torch.empty((tg.shape[0], 40), dtype=torch.uint8, device="cuda:0")
ref = None
for cl in (-1, 1):
    for pred in (False, True):
        if pred and not dp.fit_predictor(tg):
            print("no predictor:", dp._predictor_note); continue
        launch = dp.plan(tg, out=out, info_out=info, chain_len=cl, predictor=pred)
        wall, ms = bench.time_launches(launch, 20, 3, torch.device("cuda:0"))
        nfev, ok = bench.info_summary(info)
        if ref is None: ref = out.clone()
        print(f"chain_len={cl} predictor={pred}: {tg.shape[0]/wall:.4g} solves/s kernel {ms:.4f} ms evals {nfev:.3f} ok {ok} vs first {float((out-ref).abs().max()):.1e}")
