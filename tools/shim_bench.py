#!/usr/bin/env python3
"""Throughput of the camber-shim setup kernel: G perturbed geometries x random setup thickness."""
import os, sys, time
import numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.input import build_suspension
from open_kinematics_amd.shims import camber_shim_setup, shim_roles

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "shims_dw_rocker.npz"), allow_pickle=False)
sus = build_suspension(yaml.safe_load(str(g["geometry_yaml"])))
keys = list(sus.hardpoints)
names = [str(n) for n in g["names"]]
rows = [names.index(k.name.lower()) for k in keys]
rng = np.random.default_rng(0)
for n_geo in (4096, 65536, 1048576):
    base = torch.as_tensor(g["authored"][rows][None] + rng.normal(0, 0.5, size=(n_geo, len(rows), 3)), device="cuda:0").contiguous()
    shim = torch.as_tensor(np.stack([sus.camber_shim.row(30.0)] * n_geo), device="cuda:0")
    shim[:, 10] = torch.as_tensor(rng.uniform(18, 44, n_geo), device="cuda:0")
    roles = shim_roles(sus, keys)
    times = []
    for _ in range(4):
        table = base.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, info = camber_shim_setup(roles, table, shim, check=False)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    rec = info.cpu().numpy().view(np.dtype([("a", "<f8", 5), ("converged", "<i4"), ("iterations", "<i4")])).reshape(-1)
    print(f"G={n_geo:8d}  {min(times)*1e3:8.3f} ms  {n_geo/min(times):.3e} setups/s  mean iterations {rec['iterations'].mean():.2f}  all converged {bool((rec['converged']==1).all())}")
