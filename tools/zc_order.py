import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
dev = torch.device("cuda", 0)
program, targets = bump_sweep_problem(16384)
dp = DeviceProgram(program, dev)
kw = dict(chain_len=-1, predictor=False)
order = sys.argv[1]
early = bench.zero_copy_buffers(program, targets) if "Z" in order else None   # allocated before any other leg runs
for ch in order:
    if ch == "Z" and order.index("Z") != 0:
        r = bench.measure_e2e_zero_copy(dp, targets, dev, 200, kw, buffers=early); print("zero_copy (early buffers)", round(r["value"]/1e6,1), round(r["ms_per_sweep"]*1e3,1))
    if ch == "z":
        r = bench.measure_e2e_zero_copy(dp, targets, dev, 200, kw); print("zero_copy", round(r["value"]/1e6,1), round(r["ms_per_sweep"]*1e3,1))
    if ch == "c":
        r = bench.measure_e2e_compact(dp, targets, dev, 200, kw); print("compact", round(r["value"]/1e6,1))
    if ch == "e":
        r = bench.measure_e2e(dp, targets, dev, 50, kw); print("e2e", round(r["value"]/1e6,1))
    if ch == "m":
        r = bench.measure_with_model(program, torch.as_tensor(targets, device=dev), dev, 200, 10); print("with_model", round(r["value"]/1e6,1))
    if ch == "o":
        r = bench.measure_one_shot(program, torch.as_tensor(targets, device=dev), dev, 0.021); print("one_shot", round(r["value_first_launch"]/1e6,1))
    if ch == "p":
        r = bench.measure_pipelined(dp, torch.as_tensor(targets, device=dev), dev, 2000); print("pipelined", round(r["value"]/1e6,1))
