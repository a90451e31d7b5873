"""
GPU parity of the lane kernel (one lane per problem, 64 problems per wavefront, generated per program by
okx_lanegen.cpp): its residuals / J^T J / J^T r / LDL^T step against the oracle for every constraint class, and its
solves against the oracle, the quad kernel and the reference goldens - independent solves and chains, own geometry and
per-geometry tables, batches that do not fill the last wavefront.  Everything goes through the C-ABI.
"""

import numpy as np
import pytest
import torch

from conftest import gpu_available

pytestmark = pytest.mark.gpu

LANE_PROGRAMS = ["c1_dw_corner", "c4_macpherson_grid", "u_dw_corner", "u_macpherson", "rows_all_classes"]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _dp(program):
    from open_kinematics_amd.batch import DeviceProgram

    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad", f"quad kernel not loaded: {dp.kernel_note}"
    assert dp.lane_threshold > 0, f"lane kernel not loaded: {dp.lane_note}"
    return dp


@pytest.mark.parametrize("name", LANE_PROGRAMS)
@pytest.mark.parametrize("mode", ["pinned", "softnorm"])
def test_lane_normal_equations_and_step_match_the_oracle(golden, name, mode):
    """R1 / R1b for the lane kernel's generated code: r, J^T J, J^T r and the damped LDL^T step at seeded points
    (`rows_all_classes`: one row of each of the reference's 13 constraint classes)."""
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    program = program.with_line_mode(mode)
    dp = _dp(program)
    x, t = arrays["eval_x"], arrays["eval_targets"]
    r_o, jac_o = Oracle(program).eval(x, t)
    ata_o = np.einsum("bij,bik->bjk", jac_o, jac_o)
    lam = 1e-6 * float(np.max(np.diagonal(ata_o, axis1=1, axis2=2)))
    r, ata, atr, dx = [v.cpu().numpy() for v in dp.quad_eval(x, t, lam, lane=True)]
    atr_o = np.einsum("bij,bi->bj", jac_o, r_o)
    assert np.all(np.abs(r - r_o) <= 2.5e-13 + 1e-13 * np.abs(r_o))
    assert np.max(np.abs(ata - ata_o)) <= 1e-11 * max(1.0, np.abs(ata_o).max())
    assert np.max(np.abs(atr - atr_o)) <= 1e-11 * max(1.0, np.abs(atr_o).max())
    n = program.n_vars
    dx_o = np.stack([-np.linalg.solve(ata[k] + lam * np.eye(n), atr[k]) for k in range(len(x))])
    assert np.max(np.abs(dx - dx_o)) <= 1e-9 * max(1.0, np.abs(dx_o).max())
    # and the two generated kernels against each other
    rq, ataq, atrq, dxq = [v.cpu().numpy() for v in dp.quad_eval(x, t, lam)]
    assert np.all(np.abs(r - rq) <= 1e-12 + 1e-14 * np.abs(rq)) and np.max(np.abs(atr - atrq)) <= 1e-10 * max(1.0, np.abs(atrq).max())
    assert np.max(np.abs(dx - dxq)) <= 1e-9 * max(1.0, np.abs(dxq).max())


@pytest.mark.parametrize("name", ["c1_dw_corner", "c2_dw_subset", "c4_macpherson_grid", "e2e_sweep"])
def test_lane_solve_matches_oracle_quad_kernel_and_reference(golden, name):
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    t = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    lane = dp.solve(t, kernel="lane", predictor=False)
    quad = dp.solve(t, kernel="quad", predictor=False)
    torch.cuda.synchronize()
    info = lane.info()
    assert np.all((info["flags"] & 7) == 1)
    assert info["iterations"].max() <= 10
    pos = lane.positions.cpu().numpy()
    # same algorithm, same evaluation points, other summation order
    assert np.max(np.abs(pos - quad.positions.cpu().numpy())) <= 1e-10
    assert np.max(np.abs(info["nfev"] - quad.info()["nfev"])) <= 1
    sub = slice(None, None, max(1, t.shape[0] // 64))
    orc = Oracle(pinned).sweep(arrays["targets_abs"][sub], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert np.max(np.abs(pos[sub] - orc.positions)) <= 1e-9  # north-star tolerance (mm)
    assert np.max(np.abs(pos - arrays["ref_tight_pos"])) <= 6e-8  # reference's own floor
    assert np.max(np.abs(info["max_residual"] - arrays["ref_tight_maxres"])) <= 1e-8


@pytest.mark.parametrize("name", ["u_dw_corner", "u_macpherson"])
def test_lane_solve_unsteered_matches_reference_to_1e9(golden, name):
    arrays, program = golden(name)
    dp = _dp(program)
    res = dp.solve(torch.as_tensor(arrays["targets_abs"], device="cuda:0"), kernel="lane")
    torch.cuda.synchronize()
    assert np.all((res.info()["flags"] & 7) == 1)
    pos = res.positions.cpu().numpy()
    assert np.max(np.abs(pos - arrays["ref_tight_pos"])) <= 1e-9
    assert np.max(np.abs(pos - arrays["ref_default_pos"])) <= 1e-9


@pytest.mark.parametrize("shared_first_step", [True, False])
@pytest.mark.parametrize("chain_len", [-1, 1, 2, 5, 64, 65, 101, 4096])
def test_lane_chains_and_ragged_batches(golden, chain_len, shared_first_step):
    """Warm-started chains with extrapolation, independent solves, batches that do not fill the last wavefront."""
    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    t_all = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    ref = dp.solve(t_all, kernel="quad", chain_len=1).positions
    for b in (1, 3, 63, 64, 65, 101):
        guard = torch.full((b + 1, pinned.n_out, 3), -7.0, dtype=torch.float64, device="cuda:0")
        res = dp.solve(t_all[:b], kernel="lane", chain_len=chain_len, out=guard[:b], shared_first_step=shared_first_step)
        torch.cuda.synchronize()
        assert np.all((res.info()["flags"] & 7) == 1)
        assert float((res.positions - ref[:b]).abs().max()) <= 1e-9
        assert float((guard[b] + 7.0).abs().max()) == 0.0, "wrote past the batch"
    chained = dp.solve(t_all, kernel="lane", chain=True).info()
    assert chained["nfev"].mean() < dp.solve(t_all, kernel="lane", chain_len=1, predictor=False).info()["nfev"].mean()


@pytest.mark.parametrize("steps_sub", [None, 37])
def test_lane_ensemble_uses_per_geometry_tables(golden, steps_sub):
    """Per-geometry tables: a wave unit never straddles two geometries, also when the steps per geometry are not a
    multiple of 64 (the last wave unit of every geometry is partly empty)."""
    arrays, program = golden("c5_ensemble")
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    hp = torch.as_tensor(arrays["hardpoints"], device="cuda:0")
    gpos, gparam = dp.rebind(hp)
    targets = arrays["targets_abs"] if steps_sub is None else arrays["targets_abs"][:, :steps_sub]
    g, s = targets.shape[:2]
    t = torch.as_tensor(np.ascontiguousarray(targets).reshape(g * s, -1), device="cuda:0")
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s)
    lane = dp.solve(t, kernel="lane", **kw)
    quad = dp.solve(t, kernel="quad", **kw)
    torch.cuda.synchronize()
    assert np.all((lane.info()["flags"] & 7) == 1)
    assert float((lane.positions - quad.positions).abs().max()) <= 1e-10
    assert np.max(np.abs(lane.info()["nfev"] - quad.info()["nfev"])) <= 1
    ref = arrays["ref_tight_pos"] if steps_sub is None else arrays["ref_tight_pos"][:, :steps_sub]
    assert np.max(np.abs(lane.positions.cpu().numpy().reshape(g, s, -1, 3) - ref)) <= 6e-8
    for cl in (2, 4, 100):
        res = dp.solve(t, kernel="lane", chain_len=cl, **kw)
        assert np.all((res.info()["flags"] & 7) == 1)
        assert float((res.positions - lane.positions).abs().max()) <= 1e-9


def test_lane_reports_infeasible_targets_like_the_quad_kernel(golden):
    """An unreachable target stops on the ftol test with the residual flag set (solver.py:732-747)."""
    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    dp = _dp(pinned)
    t = arrays["targets_abs"][:16].copy()
    t[:, 1] += 2000.0  # wheel centre 2 m above anything the links allow
    tt = torch.as_tensor(t, device="cuda:0")
    lane = dp.solve(tt, kernel="lane").info()
    quad = dp.solve(tt, kernel="quad").info()
    assert np.all(lane["flags"] & 2) and np.all((lane["flags"] & 2) == (quad["flags"] & 2))
    assert np.max(np.abs(lane["max_residual"] - quad["max_residual"])) <= 1e-6 * np.max(quad["max_residual"])


def test_auto_selection_uses_the_lane_kernel_for_batches_that_fill_the_chip(golden):
    """From lane_threshold problems on (one more than a single round of the quad kernel holds) auto selection takes the
    lane kernel's independent-solve body; below it the quad kernel.  Same answers either way.  An ensemble with a handful
    of steps per geometry stays with the quad kernel: a lane wave unit holds problems of one geometry."""
    from open_kinematics_amd.workloads import macpherson_grid_problem

    program, targets = macpherson_grid_problem(272, 272)  # 73984 problems: not a multiple of 64 x anything neat
    dp = _dp(program)
    assert dp.lane_bodies & 1, dp.lane_note
    assert targets.shape[0] >= dp.lane_threshold
    t = torch.as_tensor(targets, device="cuda:0")
    auto = dp.solve(t, chain_len=1, predictor=False)
    quad = dp.solve(t, chain_len=1, predictor=False, kernel="quad")
    lane = dp.solve(t, chain_len=1, predictor=False, kernel="lane")
    torch.cuda.synchronize()
    assert np.all((auto.info()["flags"] & 7) == 1)
    assert float((auto.positions - quad.positions).abs().max()) <= 1e-10
    assert torch.equal(auto.positions, lane.positions), "auto did not take the lane kernel"
    small = dp.solve(t[:4096], chain_len=1, predictor=False)
    assert torch.equal(small.positions, dp.solve(t[:4096], chain_len=1, predictor=False, kernel="quad").positions)
    assert dp.lane_threshold == 4 * torch.cuda.get_device_properties(0).multi_processor_count * 16 + 1
    mid = t[: dp.lane_threshold + 1000]   # a second round of the quad kernel, a part-filled round of the lane kernel
    assert torch.equal(dp.solve(mid, chain_len=1, predictor=False).positions, dp.solve(mid, chain_len=1, predictor=False, kernel="lane").positions)
    # 4096 geometries x 8 steps: 32768 problems, but 8 of 64 lanes per wave unit -> the quad kernel
    from open_kinematics_amd.workloads import ensemble_problem

    eprog, table, rel = ensemble_problem(4096, 8)
    edp = _dp(eprog)
    gpos, gparam = edp.rebind(torch.as_tensor(table, device="cuda:0"))
    et = edp.ensemble_targets(gpos, rel)
    kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=8, chain_len=1, predictor=False)
    assert torch.equal(edp.solve(et, **kw).positions, edp.solve(et, kernel="quad", **kw).positions)


def test_chunked_ensemble_keeps_the_bits_of_the_single_launch(monkeypatch):
    """dist.ShardedEnsemble cuts a rank's shard into chunks; auto selection goes by the problem count, so the ensemble asks
    okx_plan_launch ONCE for the whole batch and forces every chunk to that kernel family and chain length: a 8192-problem
    chunk of a lane-sized ensemble runs the lane kernel too.  Rank 0 of a world of 4, alone (the exchange stubbed, the
    peers' rows put in place by hand): pre-bound solve launches, the expands of a chunk as one replayed HIP graph."""
    import open_kinematics_amd.dist as okd
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import ensemble_problem

    program, table, rel = ensemble_problem(128, 256)
    dp = DeviceProgram(program, "cuda:0")
    table = torch.as_tensor(table, device="cuda:0")
    n = 128 * 256
    assert dp.plan_launch(n, steps_per_geometry=256, geometry_tables=True, chain_len=1, predictor=False) == ("lane", 1)
    assert dp.plan_launch(n // 4, steps_per_geometry=256, geometry_tables=True, chain_len=1, predictor=False)[0] == "quad"
    assert dp.plan_launch(n, steps_per_geometry=256, geometry_tables=True, kernel="quad", chain_len=7)== ("quad", 7)
    whole = okd.ShardedEnsemble(dp, table, rel, 256, chunks=1, direct=False, chain_len=1, predictor=False)
    ref = whole.step().clone()
    alone = okd.ShardedEnsemble(dp, table, rel, 256, chain_len=1, predictor=False)  # one rank: the solves write the records
    assert alone.direct and alone.free_full is None and torch.equal(alone.step(), ref)
    coords, info = whole.free_full.clone(), whole.info_full.clone()
    torch.cuda.synchronize()

    class RankZero(okd.ShardedEnsemble):
        def _exchange_chunk(self, k):
            return []

    monkeypatch.setattr(okd, "_world", lambda group: (4, 0))
    for chunks, kind in ((4, "full"), (2, "status")):
        pipe = RankZero(dp, table, rel, 256, chunks=chunks, info=kind, chain_len=1, predictor=False)
        assert pipe.solve_kw["kernel"] == "lane" and pipe.chunks == chunks
        lo, hi = pipe.geometry_range
        own = slice(lo * 256, hi * 256)
        for step in range(3):  # plain launches, the capture, a replay
            pipe.free_full.copy_(coords)
            pipe.free_full[own] = 0.0  # what this rank has to produce itself
            got = pipe.step()
            torch.cuda.synchronize()
            assert torch.equal(pipe.free_full, coords), (chunks, step)
            assert torch.equal(got, ref), (chunks, step)
        assert all(isinstance(g, torch.cuda.CUDAGraph) for g in pipe._expand_graphs.values()) and len(pipe._expand_graphs) == chunks
        if kind == "status":
            assert torch.equal(pipe.info_local, info[own]) and torch.equal(pipe.status_full[own], info[own, 32])
        else:
            assert torch.equal(pipe.info_full[own], info[own])


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid"])
def test_a_record_does_not_depend_on_the_kernel_that_wrote_it(golden, name):
    """The lane kernel's records == okx_expand_positions_batch of its free coordinates, bit for bit (the final state's
    derived points are evaluated in the quad kernels' order of operations): one rank writing records and N ranks
    gathering coordinates and expanding them hold the same ensemble."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden(name)
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    t = np.concatenate([arrays["targets_abs"]] * 3)[:300]
    for kernel in ("lane", "quad"):
        rec = dp.solve(t, kernel=kernel, chain_len=1, predictor=False)
        free = dp.solve(t, kernel=kernel, chain_len=1, predictor=False, output="free")
        torch.cuda.synchronize()
        assert torch.equal(rec.positions[:, dp.free_out_index], free.free), kernel
        assert torch.equal(dp.expand(free.free), rec.positions), kernel
