import sys, numpy as np, torch
sys.path.insert(0, 'tests')
from conftest import load_golden
from open_kinematics_amd.batch import DeviceProgram
for name in ["t_axle_t_bar_bump", "t_axle_t_bar_roll", "t_axle_heave_link"]:
    arrays, program = load_golden(name)
    pinned = program.with_line_mode("pinned")
    dp = DeviceProgram(pinned, "cuda:0")
    print(name, "n", program.n_vars, "m", program.n_rows, "kernel", dp.kernel, "|", dp.kernel_note, "| lane:", dp.lane_note)
    t = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    for kern in ("wave", "auto"):
        try:
            res = dp.solve(t, kernel=kern)
        except Exception as e:
            print("  ", kern, "ERR", e); continue
        torch.cuda.synchronize()
        info = res.info()
        pos = res.positions.cpu().numpy()
        print("  ", kern, "conv", int(((info["flags"] & 7) == 1).sum()), "/", len(info), "nfev", info["nfev"].mean(), "iters max", info["iterations"].max(),
              "max|pos-ref_tight|", np.max(np.abs(pos - arrays["ref_tight_pos"])), "maxres", info["max_residual"].max(), "flags", np.unique(info["flags"]))
