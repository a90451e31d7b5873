#!/usr/bin/env python3
"""The C2 cold solve of the SAME 16 problems replicated over 1 ... 1024 wavefronts: how much of the sweep time is one
wavefront's instruction stream and how much the chip adds (dispatch of 1024 workgroups, 6.4 MB of records, shared fetch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd import workloads as W
dev = torch.device("cuda:0")
p, t = W.bump_sweep_problem(16384)
dp = DeviceProgram(p, dev)
import numpy as np
# every launch solves the SAME 16 far problems replicated, so all waves do identical work
tt = np.tile(t[:16], (1024, 1))
tg = torch.as_tensor(tt, device=dev)
out = torch.empty((tg.shape[0], p.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((tg.shape[0], 40), dtype=torch.uint8, device=dev)
for n in (16, 256, 1024, 2048, 4096, 8192, 12288, 16384):
    launch = dp.plan(tg[:n], out=out[:n], info_out=info[:n], chain_len=1, predictor=False)
    wall, ms = bench.time_launches(launch, 200, 10, dev)
    nfev, ok = bench.info_summary(info[:n])
    print(f"{n:6d} problems ({n//16:5d} wavefronts): {1e3*ms:7.2f} us, evaluations {nfev:.2f}", flush=True)
