import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
dev = torch.device("cuda", 0)
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, dev)
print(json.dumps(bench.measure_e2e_compact(dp, targets, dev, 400, dict(chain_len=-1, predictor=False))))
print(json.dumps(bench.measure_e2e(dp, targets, dev, 50, dict(chain_len=-1, predictor=False))))
