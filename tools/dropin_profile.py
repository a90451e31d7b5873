#!/usr/bin/env python3
"""Where a drop-in solve_sweep call spends its time (host profile + kernel time)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yaml, torch
from open_kinematics_amd.input import build_sweep, load_geometry
from open_kinematics_amd.sweep import solve_sweep
from open_kinematics_amd.workloads import geometry_path
sus = load_geometry(geometry_path("geometry.yaml"))
m = yaml.safe_load(open(geometry_path("bump_sweep.yaml"))); m["steps"] = 101
sweep = build_sweep(m, sus)
for _ in range(3): solve_sweep(sus, sweep)
t0 = time.perf_counter()
for _ in range(50): solve_sweep(sus, sweep)
print("per call ms", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): solve_sweep(sus, sweep)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
