#!/usr/bin/env python3
"""One BASELINE configuration at full size, K launches of the solve kernel, one JSON line (kernel time from HIP events).
What tools/profile_run.sh wraps in rocprofv3 for the configurations bench.py's headline does not cover.
   python3 tools/profile_config.py c3|c4|c5|c2 cold|chained [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from open_kinematics_amd import workloads as W

name, mode = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
make = {"c2": lambda: W.bump_sweep_problem(16384), "c3": lambda: W.axle_grid_problem(256, 256),
        "c4": lambda: W.macpherson_grid_problem(512, 512), "c5": lambda: W.ensemble_problem(4096, 256)}[name]
torch.cuda.set_device(0)
res = bench.measure_config(name, make, torch.device("cuda", 0), steps, 3, modes=(mode,))
res["mode"] = mode
res["steps"] = steps
print(json.dumps(res))
