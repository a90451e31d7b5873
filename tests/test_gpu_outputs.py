"""
okx_solve_opts.output on the GPU: the compact forms of a solve's result - the solved free points alone ([B][n_free][3]:
what a PCIe link or an all-gather wants to carry) or nothing but the info records - against the full records, for the
quad kernel (single and pair mode) and the lane kernel, independent solves and chains.
"""

import numpy as np
import pytest
import torch

from conftest import gpu_available

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _cases():
    from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem

    return {"dw": lambda: bump_sweep_problem(1000), "mac": lambda: macpherson_grid_problem(30, 31),
            "axle": lambda: axle_grid_problem(12, 13)}


@pytest.mark.parametrize("kernel", ["quad", "lane"])
@pytest.mark.parametrize("chain_len", [1, 7, -1])
@pytest.mark.parametrize("case", ["dw", "mac", "axle"])
def test_free_coordinates_and_none_match_the_full_records(case, chain_len, kernel):
    from open_kinematics_amd.batch import DeviceProgram

    program, targets = _cases()[case]()
    dp = DeviceProgram(program, "cuda:0")
    if kernel == "lane" and dp.lane_threshold < 0:
        pytest.skip("no lane kernel for this program")
    t = torch.as_tensor(targets, device="cuda:0")
    kw = dict(chain_len=chain_len, kernel=kernel, predictor=False)
    full = dp.solve(t, **kw)
    free = dp.solve(t, output="free", **kw)
    none = dp.solve(t, output="none", **kw)
    torch.cuda.synchronize()
    assert full.free is None and free.positions is None and none.positions is None and none.free is None
    assert free.free.shape == (t.shape[0], program.n_free, 3)
    assert np.all((full.info()["flags"] & 7) == 1)
    # the same solve: bit-identical free points, identical info records
    assert torch.equal(free.free, full.positions.index_select(1, dp.free_out_index))
    for other in (free, none):
        a, b = full.info(), other.info()
        assert np.array_equal(a["flags"], b["flags"]) and np.array_equal(a["nfev"], b["nfev"])
        assert np.array_equal(a["max_residual"], b["max_residual"])
    if case != "axle":  # (the expand kernel belongs to single-mode programs)
        rebuilt = dp.expand(free.free)
        if kernel == "quad":  # the expand kernel IS the quad kernel's final-state code: bit-identical records
            assert torch.equal(rebuilt, full.positions)
        else:                 # the lane kernel evaluates the derived points with its own (scalar) operation order
            assert float((rebuilt - full.positions).abs().max()) <= 1e-12
            assert torch.equal(rebuilt.index_select(1, dp.free_out_index), free.free)
    # a caller's buffer of the right shape is used, one of the wrong shape refused
    mine = torch.full((t.shape[0] + 1, program.n_free, 3), -3.0, dtype=torch.float64, device="cuda:0")
    res = dp.solve(t, output="free", out=mine[:-1], **kw)
    torch.cuda.synchronize()
    assert res.free.data_ptr() == mine.data_ptr() and torch.equal(res.free, free.free)
    assert float((mine[-1] + 3.0).abs().max()) == 0.0, "wrote past the batch"
    with pytest.raises(ValueError):
        dp.solve(t, output="free", out=torch.empty((t.shape[0], program.n_out, 3), dtype=torch.float64, device="cuda:0"), **kw)


def test_compact_outputs_with_per_geometry_tables(golden):
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c5_ensemble")
    pinned = program.with_line_mode("pinned")
    dp = DeviceProgram(pinned, "cuda:0")
    gpos, gparam = dp.rebind(torch.as_tensor(arrays["hardpoints"], device="cuda:0"))
    g, s = arrays["targets_abs"].shape[:2]
    t = torch.as_tensor(arrays["targets_abs"].reshape(g * s, -1), device="cuda:0")
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s)
    for kernel in ("quad", "lane"):
        full = dp.solve(t, kernel=kernel, **kw)
        free = dp.solve(t, kernel=kernel, output="free", **kw)
        torch.cuda.synchronize()
        assert torch.equal(free.free, full.positions.index_select(1, dp.free_out_index))
        rebuilt = dp.expand(free.free, geom_pos=gpos, steps_per_geometry=s)
        assert torch.equal(rebuilt, full.positions) if kernel == "quad" else float((rebuilt - full.positions).abs().max()) <= 1e-12


def test_the_interpreter_kernels_only_write_records(golden):
    from open_kinematics_amd import _lib
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c1_dw_corner")
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    with pytest.raises(ValueError, match="needs a generated kernel"):
        dp.solve(arrays["targets_abs"], kernel="single", output="free")
    assert dp.solve(arrays["targets_abs"], kernel="single").positions is not None


def test_pinned_host_buffers_are_accepted_as_device_accessible_pointers(golden):
    """okx.h: d_* pointers may be pinned host memory.  Targets read from and compact outputs stored straight into the
    caller's pinned buffers hold the same bits as a launch on HBM buffers (bench.py e2e.zero_copy)."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c2_dw_subset")
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    n = arrays["targets_abs"].shape[0]
    h_t = torch.as_tensor(arrays["targets_abs"]).pin_memory()
    for mode, shape in (("free", (n, program.n_free, 3)), ("records", (n, program.n_out, 3))):
        h_out = torch.full(shape, -7.0, dtype=torch.float64).pin_memory()
        h_info = torch.zeros((n, 40), dtype=torch.uint8).pin_memory()
        dp.solve(h_t, out=h_out, info_out=h_info, output=mode, chain_len=-1, zero_copy=True)
        torch.cuda.synchronize()
        ref = dp.solve(torch.as_tensor(arrays["targets_abs"], device="cuda:0"), output=mode, chain_len=-1)
        torch.cuda.synchronize()
        want = ref.free if mode == "free" else ref.positions
        assert np.array_equal(h_out.numpy(), want.cpu().numpy())
        assert np.array_equal(h_info.numpy(), ref.info_raw.cpu().numpy())


def test_pinned_inputs_are_snapshotted_unless_zero_copy_is_asked_for(golden):
    """pinned host tensors passed to rebind / eval / solve WITHOUT zero_copy are copied to the device like any
    other host data (no pointer into host memory reaches a kernel), pinned OUTPUT buffers are refused without the flag, and
    pageable host buffers always."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c2_dw_subset")
    program = program.with_line_mode("pinned")
    dp = DeviceProgram(program, "cuda:0")
    t_host = torch.as_tensor(arrays["targets_abs"]).pin_memory()
    t_dev = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    # rebind with a pinned hardpoint table: outputs live on the device and match the device-input call
    hp = torch.as_tensor(np.repeat(program.design_pos[None], 3, axis=0)).pin_memory()
    gpos, gparam = dp.rebind(hp)
    gpos_d, gparam_d = dp.rebind(hp.to("cuda:0"))
    assert gpos.is_cuda and gparam.is_cuda and torch.equal(gpos, gpos_d) and torch.equal(gparam, gparam_d)
    # eval with pinned x and targets
    x = torch.as_tensor(np.repeat(program.design_pos[program.free_point].reshape(1, -1), t_host.shape[0], axis=0)).pin_memory()
    r_pinned = dp.eval(x, t_host)
    r_dev = dp.eval(x.to("cuda:0"), t_dev)
    assert all(a.is_cuda and torch.equal(a, b) for a, b in zip(r_pinned, r_dev))
    # solve: a pinned input is snapshotted - overwriting it right after the call does not change the answer
    scratch = t_host.clone().pin_memory()
    res = dp.solve(scratch, chain_len=1)
    scratch.zero_()
    torch.cuda.synchronize()
    ref = dp.solve(t_dev, chain_len=1)
    assert torch.equal(res.positions, ref.positions)
    # output buffers on the host need the flag; pageable ones are never accepted
    n = t_host.shape[0]
    h_out = torch.empty((n, program.n_out, 3), dtype=torch.float64).pin_memory()
    h_info = torch.zeros((n, 40), dtype=torch.uint8).pin_memory()
    with pytest.raises(ValueError, match="zero_copy=True"):
        dp.solve(t_dev, out=h_out, info_out=h_info)
    with pytest.raises(ValueError, match="pinned host tensor"):
        dp.solve(t_dev, out=torch.empty((n, program.n_out, 3), dtype=torch.float64), info_out=h_info, zero_copy=True)
    dp.close()


def test_a_non_finite_lambda0_is_refused(golden):
    """every distinct lambda0 owns a first-step table; NaN never matches a cached one and used to allocate a table
    per launch.  The launch is refused instead, and the table list of a program is capped."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c2_dw_subset")
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    t = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    for bad in (float("nan"), float("inf"), -1.0):
        with pytest.raises(ValueError, match="lambda0"):
            dp.solve(t, lambda0=bad)
    ref = dp.solve(t, chain_len=1).positions
    for k in range(24):  # more distinct values than a program keeps tables for: the later ones solve without a table
        res = dp.solve(t, chain_len=1, lambda0=1e-6 * (1.0 + 0.01 * k))
        assert res.accepted(res.info()).all()
        assert float((res.positions - ref).abs().max()) <= 1e-9
    dp.close()
