"""C3 (256 x 256 axle grid): the evaluated solve of the composed axle as ONE launch against the separate launches.
    python tools/axle_evaluated_rate.py [n_heave n_roll]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import axle_grid_problem, geometry_path

    nh, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
    device = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device))
    program, targets = axle_grid_problem(nh, nr)
    dp = DeviceProgram(program, device)
    t = torch.as_tensor(targets, device=device)
    n = t.shape[0]
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    if "--only" in sys.argv:  # the one fused launch alone (what tools/profile_run.sh wraps in rocprofv3)
        from open_kinematics_amd.input import load_geometry
        from open_kinematics_amd.metrics import axle_evaluation_roles

        roles, _, _ = axle_evaluation_roles(load_geometry(geometry_path("axle_geometry_rocker.yaml")), program)
        dp.enable_evaluation(roles)
        evb = torch.empty((n, 1 + program.n_targets, dp.eval_columns), dtype=torch.float64, device=device)
        launch = dp.plan_evaluated(t, info_out=info, eval_out=evb, output="none", chain_len=1, predictor=False)
        _, ms = bench.time_launches(launch, 20, 5, device)
        print(json.dumps({"workload": "C3 rocker + U-bar axle, 256x256 heave x roll grid, evaluated in one launch (output = none)",
                          "states": n, "roofline": {"kernel_ms": ms}, "states_per_s": n / ms * 1e3,
                          "algorithmic_bytes_per_state": 8 * program.n_targets + 8 * dp.eval_columns * (1 + program.n_targets) + 16}))
        return
    res = bench.measure_evaluated_axle(dp, geometry_path("axle_geometry_rocker.yaml"), t, out, info, {}, device, 20, 5)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
