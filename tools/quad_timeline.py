#!/usr/bin/env python3
"""C2 cold sweep: the cold body against the general body (same answers? time per launch), and - from a build generated with
OKX_DEV=quad_timeline - where a wavefront's cycles go (shader-clock stamps of every wavefront: entry, first step in hand, top of
each LM pass, passes done, records stored, end).
   OKX_DEV=quad_timeline OKX_KERNEL_CACHE=build/kc_tl python3 tools/quad_timeline.py [n_problems | c3]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem

what = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "c2"
dev = torch.device("cuda:0")
if what == "c3":   # one round of the pair-mode kernel: 8 problems per wavefront, 1024 wavefronts
    from open_kinematics_amd.workloads import axle_grid_problem
    program, targets = axle_grid_problem(128, 64)
    n, per_wave = targets.shape[0], 8
else:
    n, per_wave = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 16
    program, targets = bump_sweep_problem(n)
dp = DeviceProgram(program, dev)
t = torch.as_tensor(targets, device=dev)
out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=dev)
info = torch.empty((n, 40), dtype=torch.uint8, device=dev)


def run(label):
    launch = dp.plan(t, out=out, info_out=info, chain_len=1, predictor=False)
    for _ in range(300):
        launch()
    wall, ms = bench.time_launches(launch, 2000, 100, dev)
    torch.cuda.synchronize()
    pos = out.cpu().numpy().copy()
    inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=[("max_residual", "f8"), ("cost", "f8"), ("last_step", "f8"), ("iterations", "i4"),
                                                              ("nfev", "i4"), ("flags", "i4"), ("reserved", "i4")])
    print(f"{label:10s}: {1e3 * ms:7.2f} us per launch, nfev mean {inf['nfev'].mean():.3f}, converged {np.all((inf['flags'] & 7) == 1)}")
    return pos, inf


if "quad_timeline" in os.environ.get("OKX_DEV", ""):
    waves = (n + per_wave - 1) // per_wave
    tr = torch.zeros((2 * waves, 16), dtype=torch.float64, device=dev)
    dp.lib.okx_debug_quad_trace(dp._handle, C.c_void_p(tr.data_ptr()), -1)
    launch = dp.plan(t, out=out, info_out=info, chain_len=1, predictor=False)
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    tr.zero_()
    launch()
    torch.cuda.synchronize()
    both = tr.cpu().numpy()
    a, sec = both[:waves], both[waves:]
    dp.lib.okx_debug_quad_trace(dp._handle, None, -1)
    t0 = a[:, 0].min()
    names = ["entry", "first step in hand"] + [f"pass {k} top" for k in range(1, 11)] + ["(pass overflow)", "passes done", "records stored", "end"]
    # (the shader clocks of the eight XCDs have different bases: only differences inside one wavefront mean anything)
    print("per wavefront: ticks from the previous stamp it has to this one - median / min / max over the wavefronts that have it")
    order = [k for k in range(16) if (a[:, k] > 0).any()]
    for k in order[1:]:
        have = a[:, k] > 0
        prev = np.zeros(have.sum())
        for j in order[:order.index(k)]:
            col = a[have, j]
            prev = np.where(col > 0, col, prev)
        d = a[have, k] - prev
        print(f"  {k:2d} {names[k]:22s} n={have.sum():5d}  {np.median(d):9.0f} {d.min():9.0f} {d.max():9.0f}")
    # the 100 MHz real-time counter (one time base for the whole device): when the wavefronts start and end within the launch
    rt0, rt1 = sec[:, 0], sec[:, 1]
    if (rt0 > 0).all():
        base = rt0.min()
        print(f"real time (10 ns ticks): wavefront starts spread over {rt0.max() - base:.0f} ticks (median start {np.median(rt0 - base):.0f}), "
              f"ends {np.median(rt1 - base):.0f} median / {(rt1 - base).max():.0f} last; lifetime median {np.median(rt1 - rt0):.0f} ticks "
              f"= {np.median(a[:, 15] - a[:, 0]) / np.median(rt1 - rt0) / 10:.3f} shader cycles per ns")
    per_wave = a[:, 15] - a[:, 0]
    print(f"wavefront lifetime entry -> end: median {np.median(per_wave):.0f}, min {per_wave.min():.0f}, max {per_wave.max():.0f} ticks;"
          f" whole launch (first entry -> last end) {(a[:, 15].max() - t0):.0f} ticks")
    # sections of the second full pass (wavefronts that ran one): deltas along top -> derived points -> rows -> LM decision ->
    # factorisation -> substitution -> end of pass
    has = sec[:, 11] > 0
    if has.any():
        order = [(8, "derived points + chain blocks"), (9, "rows: residuals, gradients, J^T r, J^T J"), (10, "LM decision"),
                 (5, "(to factorisation)"), (6, "LDL^T factorisation"), (7, "substitutions"), (11, "step norms, next-pass logic")]
        prev = a[has, 3]
        print(f"sections of the second full pass (median ticks over {has.sum()} wavefronts):")
        for k, label in order:
            cur = sec[has, k]
            print(f"  {label:44s} {np.median(cur - prev):8.0f}")
            prev = cur
        nxt = np.where(a[has, 4] > 0, a[has, 4], a[has, 13])
        print(f"  {'(to the next pass top / loop exit)':44s} {np.median(nxt - prev):8.0f}")
    passes = (a[:, 2:12] > 0).sum(axis=1)
    print("LM passes per wavefront:", dict(zip(*[x.tolist() for x in np.unique(passes, return_counts=True)])))
    sys.exit(0)

pos_c, inf_c = run("cold body")
os.environ["OKX_DEV"] = "no_cold"
pos_g, inf_g = run("general")
print(f"max |cold - general| = {np.max(np.abs(pos_c - pos_g)):.3e} mm; nfev equal: {np.array_equal(inf_c['nfev'], inf_g['nfev'])}")
