"""
GPU parity tests: the HIP path (through the C-ABI, libokx.so) against the CPU oracle on
the same seeded inputs and against the committed reference goldens.  Run on the MI355X box
with ``pytest -m gpu``.  Tolerances are the ladder of DESIGN.md §4 / SURVEY.md §8c:

  R1  residuals / Jacobians                      <= 2.5e-13 (1 ulp of a 1 m length) / 1e-13
  R2  solved positions, well-posed problems      <= 1e-9 mm  (north_star tolerance)
      (unsteered goldens of the reference; oracle LM on the identical program)
  R3  rack-steered goldens of the reference      <= 6e-8 mm on the rack pickup (the
      reference's own convergence floor), <= 1e-8 mm on every other point
"""

import csv
import os

import numpy as np
import pytest

from conftest import ALL_ROW_CLASSES, GOLDEN, STEERED, UNSTEERED

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _device_program(program):
    from open_kinematics_amd.batch import DeviceProgram

    return DeviceProgram(program)


def _solve(program, targets, **kw):
    dp = _device_program(program)
    res = dp.solve(torch.as_tensor(np.ascontiguousarray(targets)), **kw)
    torch.cuda.synchronize()
    return res.positions.cpu().numpy(), res.info()


@pytest.mark.parametrize("name", STEERED + UNSTEERED + ALL_ROW_CLASSES)
@pytest.mark.parametrize("line_mode", ["softnorm", "pinned"])
def test_eval_matches_oracle_and_reference(golden, name, line_mode):
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    program = program.with_line_mode(line_mode)
    dp = _device_program(program)
    r, jac = dp.eval(arrays["eval_x"], arrays["eval_targets"])
    torch.cuda.synchronize()
    r, jac = r.cpu().numpy(), jac.cpu().numpy()
    r_o, jac_o = Oracle(program).eval(arrays["eval_x"], arrays["eval_targets"])
    # absolute floors for mm-scale rows, relative for the large ones (unscaled volumes of the coplanar row)
    assert np.all(np.abs(r - r_o) <= 2.5e-13 + 1e-13 * np.abs(r_o))
    assert np.all(np.abs(jac - jac_o) <= 1e-13 * np.maximum(1.0, np.abs(jac_o)))
    if line_mode == "softnorm":  # identical row set as the reference itself
        ref_r, ref_j = arrays["eval_r"], arrays["eval_jac"]
        assert np.all(np.abs(r - ref_r) <= 2.5e-13 + 1e-13 * np.abs(ref_r))
        assert np.all(np.abs(jac - ref_j) <= 1e-13 * np.maximum(1.0, np.abs(ref_j)))


@pytest.mark.parametrize("name", ["c1_dw_corner", "c3_axle_grid", "c4_macpherson_grid", "u_axle"])
def test_normal_equations_match_dense_product(golden, name):
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    program = program.with_line_mode("pinned")
    dp = _device_program(program)
    r, ata, atr = dp.normal_equations(arrays["eval_x"], arrays["eval_targets"])
    torch.cuda.synchronize()
    r_o, jac_o = Oracle(program).eval(arrays["eval_x"], arrays["eval_targets"])
    ata_o = np.einsum("bij,bik->bjk", jac_o, jac_o)
    atr_o = np.einsum("bij,bi->bj", jac_o, r_o)
    assert np.max(np.abs(ata.cpu().numpy() - ata_o)) <= 1e-11 * max(1.0, np.abs(ata_o).max())
    assert np.max(np.abs(atr.cpu().numpy() - atr_o)) <= 1e-11 * max(1.0, np.abs(atr_o).max())


@pytest.mark.parametrize("name", UNSTEERED)
def test_solve_unsteered_matches_reference(golden, name):
    """R2 against the reference's own outputs (no degenerate row in these problems)."""
    arrays, program = golden(name)
    pos, info = _solve(program, arrays["targets_abs"])
    assert np.all((info["flags"] & 1) == 1)
    assert np.max(np.abs(pos - arrays["ref_tight_pos"])) <= 1e-9
    assert np.max(np.abs(pos - arrays["ref_default_pos"])) <= 1e-9
    assert info["iterations"].max() <= 8


@pytest.mark.parametrize("name", STEERED)
def test_solve_steered_against_oracle_and_reference(golden, name):
    from oracle.oracle import Oracle

    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    targets = arrays["targets_abs"]
    pos, info = _solve(pinned, targets)
    assert np.all((info["flags"] & 7) == 1), "every problem converged within tolerance"
    assert info["iterations"].max() <= 10
    # R2: the oracle's MINPACK LM on the IDENTICAL (pinned) program, cold start
    sub = slice(None, None, max(1, targets.shape[0] // 64))
    orc = Oracle(pinned).sweep(targets[sub], 1e-15, 1e-15, 1e-15, warm_start=False)
    assert orc.first_failed_step == -1
    assert np.max(np.abs(pos[sub] - orc.positions)) <= 1e-9
    # R3: the reference's own (softnorm, sequential, tight) outputs
    diff = np.abs(pos - arrays["ref_tight_pos"])
    rack = [i for i, k in enumerate(program.out_point)
            if program.point_keys[k].lower_name.endswith("trackrod_inboard")]
    assert diff.max() <= 6e-8
    assert np.delete(diff, rack, axis=1).max() <= 1e-8
    assert np.max(np.abs(pos - arrays["ref_default_pos"])) <= 5e-5
    # reported max_residual uses the reference's row definitions
    assert np.max(np.abs(info["max_residual"] - arrays["ref_tight_maxres"])) <= 1e-8


@pytest.mark.parametrize("name", ["c1_dw_corner", "c3_axle_grid", "c4_macpherson_grid"])
def test_softnorm_rows_on_device_reach_the_same_point(golden, name):
    """The reference's own (degenerate) row set also runs on device - corner, composed axle (pair mode) and MacPherson;
    looser step tolerance: along the zero-gradient line row the iteration converges linearly."""
    arrays, program = golden(name)
    assert any(int(t) == 8 for t in program.row_type)  # the literal point-on-line row
    pos_s, info_s = _solve(program, arrays["targets_abs"], step_tol=1e-8, max_iter=200)
    pos_p, _ = _solve(program.with_line_mode("pinned"), arrays["targets_abs"])
    assert np.all((info_s["flags"] & 1) == 1)
    assert np.max(np.abs(pos_s - pos_p)) <= 5e-8
    assert np.max(np.abs(pos_s - arrays["ref_tight_pos"])) <= 6e-8


def test_chain_mode_follows_reference_warm_start(golden):
    """solver.py:716,774: one wavefront walks the sweep, step k starts at step k-1."""
    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    t = arrays["targets_abs"]
    pos_c, info_c = _solve(pinned, t, chain=True)
    pos_i, info_i = _solve(pinned, t, predictor=False)  # independent cold starts from the design state
    assert np.all((info_c["flags"] & 7) == 1)
    assert np.max(np.abs(pos_c - pos_i)) <= 1e-9
    assert info_c["iterations"].mean() < info_i["iterations"].mean()


@pytest.mark.parametrize("chain_len", [-1, 1, 7, 50, 101, 1000])
def test_chunked_chains_give_the_same_answers(golden, chain_len):
    """Consecutive steps grouped into warm-started chains of any length (incl. ragged tails)."""
    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    pos_i, _ = _solve(pinned, arrays["targets_abs"], chain_len=1)
    pos_c, info = _solve(pinned, arrays["targets_abs"], chain_len=chain_len)
    assert np.all((info["flags"] & 7) == 1)
    assert np.max(np.abs(pos_c - pos_i)) <= 1e-9


def test_chains_never_cross_geometries(golden):
    arrays, program = golden("c5_ensemble")
    pinned = program.with_line_mode("pinned")
    dp = _device_program(pinned)
    gpos, gparam = dp.rebind(torch.as_tensor(arrays["hardpoints"]))
    g, s = arrays["targets_abs"].shape[:2]
    t = torch.as_tensor(arrays["targets_abs"].reshape(g * s, -1))
    ref = dp.solve(t, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s, chain_len=1)
    for chain_len in (4, 9, 100):
        res = dp.solve(t, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=s, chain_len=chain_len)
        torch.cuda.synchronize()
        assert np.all((res.info()["flags"] & 7) == 1)
        assert float((res.positions - ref.positions).abs().max()) <= 1e-9


def test_e2e_golden_csv_of_the_reference(golden):
    arrays, program = golden("e2e_sweep")
    pos, info = _solve(program.with_line_mode("pinned"), arrays["targets_abs"])
    with open(os.path.join(GOLDEN, "e2e_output.csv"), "r", encoding="utf-8") as fh:
        rows = list(csv.DictReader([ln for ln in fh if not ln.strip().startswith("#")]))
    worst = 0.0
    for k, idx in enumerate(program.out_point):
        name = program.point_keys[idx].lower_name
        for a, axis in enumerate("xyz"):
            col = np.array([float(r[f"{name}_{axis}"]) for r in rows])
            worst = max(worst, float(np.max(np.abs(col - pos[:, k, a]))))
    assert worst <= 5e-5  # the reference's own tolerance here is 1e-3 (test_e2e.py:204-211)


def test_rebind_and_ensemble_solve(golden):
    """C5: per-geometry design targets on device + geometry-major batch."""
    from oracle.oracle import Oracle

    arrays, program = golden("c5_ensemble")
    pinned = program.with_line_mode("pinned")
    dp = _device_program(pinned)
    hp = torch.as_tensor(arrays["hardpoints"])
    gpos, gparam = dp.rebind(hp)
    torch.cuda.synchronize()
    assert np.max(np.abs(gpos.cpu().numpy() - arrays["design_pos"])) <= 1e-12
    orc = Oracle(pinned)
    for g in range(hp.shape[0]):
        _, rp = orc.rebind(arrays["hardpoints"][g])
        assert np.max(np.abs(gparam[g].cpu().numpy() - rp) / np.maximum(1.0, np.abs(rp))) <= 1e-14
    g, s = arrays["targets_abs"].shape[:2]
    res = dp.solve(torch.as_tensor(arrays["targets_abs"].reshape(g * s, -1)), geom_pos=gpos,
                   geom_row_param=gparam, steps_per_geometry=s)
    torch.cuda.synchronize()
    info = res.info()
    assert np.all((info["flags"] & 7) == 1)
    pos = res.positions.cpu().numpy().reshape(g, s, -1, 3)
    diff = np.abs(pos - arrays["ref_tight_pos"])
    rack = [i for i, k in enumerate(program.out_point)
            if program.point_keys[k].lower_name.endswith("trackrod_inboard")]
    assert diff.max() <= 6e-8
    assert np.delete(diff, rack, axis=2).max() <= 1e-8


def test_full_size_bump_sweep_properties(golden):
    """BASELINE config 2 at full size (16384 steps): size-independent properties."""
    arrays, program = golden("c2_dw_subset")
    pinned = program.with_line_mode("pinned")
    t0 = arrays["targets_abs"][0]
    design_z = arrays["targets_abs"][:, 1] - np.linspace(-60, 80, 16384)[arrays["subset_index"]]
    targets = np.stack([np.full(16384, t0[0]), design_z[0] + np.linspace(-60.0, 80.0, 16384)], 1)
    pos, info = _solve(pinned, targets)
    assert np.all((info["flags"] & 7) == 1)
    assert info["max_residual"].max() <= 1e-6
    names = [program.point_keys[k].lower_name for k in program.out_point]
    wc = pos[:, names.index("wheel_center")]
    assert np.max(np.abs(wc[:, 2] - targets[:, 1])) <= 1e-9          # the target row is met
    assert np.all(np.diff(wc[:, 2]) > 0)                              # monotone sweep
    # rigid links keep their design length at every step (softnorm bias ~1e-6 allowed)
    def length(a, b):
        return np.linalg.norm(pos[:, names.index(a)] - pos[:, names.index(b)], axis=1)
    for a, b in [("upper_wishbone_inboard_front", "upper_wishbone_outboard"),
                 ("lower_wishbone_inboard_rear", "lower_wishbone_outboard"),
                 ("trackrod_inboard", "trackrod_outboard"), ("axle_inboard", "axle_outboard")]:
        d = length(a, b)
        assert np.max(np.abs(d - d[0])) <= 5e-6
    # the strided subset agrees with the reference golden of exactly those steps
    sub = pos[arrays["subset_index"]]
    assert np.max(np.abs(sub - arrays["ref_tight_pos"])) <= 6e-8
    # fixed points are passed through untouched
    fixed = names.index("lower_wishbone_inboard_front")
    assert np.all(pos[:, fixed] == pos[0, fixed])


def test_infeasible_target_is_flagged_not_fatal(golden):
    arrays, program = golden("u_dw_corner")
    targets = arrays["targets_abs"][:4].copy()
    targets[2, 0] += 2000.0
    pos, info = _solve(program, targets, max_iter=100)
    assert (info["flags"][2] & 2) == 2 and info["max_residual"][2] > 1e-3
    ok = [0, 1, 3]
    assert np.all((info["flags"][ok] & 7) == 1)
    assert np.all(np.isfinite(pos))


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "c3_axle_grid"])
def test_positions_rebuilt_from_free_coordinates(golden, name):
    """okx_expand_positions_batch: the free coordinates of a solve carry its whole state (multi-GPU exchange payload)."""
    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    dp = _device_program(pinned)
    res = dp.solve(torch.as_tensor(arrays["targets_abs"]))
    free = res.positions[:, dp.free_out_index]
    assert free.shape == (res.positions.shape[0], pinned.n_free, 3)
    rebuilt = dp.expand(free)
    torch.cuda.synchronize()
    assert float((rebuilt - res.positions).abs().max()) <= 1e-11
    guard = torch.full((5, pinned.n_out, 3), -3.0, dtype=torch.float64, device=rebuilt.device)
    dp.expand(free[:4], out=guard[:4])
    assert float((guard[:4] - res.positions[:4]).abs().max()) <= 1e-11 and float((guard[4] + 3.0).abs().max()) == 0.0
    with pytest.raises(ValueError, match="out must be"):
        dp.expand(free[:4], out=guard)


@pytest.mark.parametrize("name", ["c1_dw_corner", "c3_axle_grid"])  # (generated expand kernel; the interpreter's: one thread per state)
def test_positions_rebuilt_with_per_geometry_tables(golden, name):
    arrays, program = golden(name)
    pinned = program.with_line_mode("pinned")
    dp = _device_program(pinned)
    rng = np.random.default_rng(3)
    table = np.repeat(pinned.design_pos[None], 3, axis=0)
    table[1:, pinned.free_point] += rng.normal(0.0, 0.3 if name == "c1_dw_corner" else 0.05, size=(2, pinned.n_free, 3))
    gpos, gparam = dp.rebind(torch.as_tensor(table))
    lo = 40 if arrays["targets_abs"].shape[0] >= 60 else 0
    t = torch.as_tensor(np.tile(arrays["targets_abs"][lo:lo + 20], (3, 1)))
    res = dp.solve(t, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=20)
    rebuilt = dp.expand(res.positions[:, dp.free_out_index], geom_pos=gpos, steps_per_geometry=20)
    torch.cuda.synchronize()
    assert np.all(res.accepted(res.info()))
    assert float((rebuilt - res.positions).abs().max()) <= 1e-11


def test_compact_exchange_pipeline_on_the_device(golden, monkeypatch):
    """FreeGatherPipeline with a stand-in two-rank all-gather: solve into the send buffer (output = free), gather, one expand
    of the gathered block on the GPU gives every rank's full positions."""
    import torch.distributed as dist

    from open_kinematics_amd import dist as okx_dist

    arrays, program = golden("c1_dw_corner")
    pinned = program.with_line_mode("pinned")
    dp = _device_program(pinned)
    t = torch.as_tensor(arrays["targets_abs"], device=dp.device)
    half = t.shape[0] // 2
    t = t[: 2 * half]
    ref = dp.solve(t).positions

    class Done:
        def wait(self):
            return True

    other = {}

    def fake_all_gather(full, local, group=None, async_op=False):
        # rank 0's view of a two-rank job: its own block first, the peer's (solved here too) second
        full[:half].copy_(local)
        full[half:].copy_(other["free"])
        return Done()

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    pipe = okx_dist.FreeGatherPipeline(half, pinned.n_out, pinned.n_free, dp.expand, torch.float64, dp.device)
    assert pipe.world == 2 and pipe.output == "free" and pipe.gathered[0].shape == (2 * half, pinned.n_free, 3)
    other["free"] = ref[half:, dp.free_out_index].contiguous()
    for k in range(3):
        out = pipe.begin(k)                       # the send buffer: the solve writes the free coordinates into it itself
        dp.solve(t[:half], out=out, output=pipe.output)
        pipe.submit(k)
    full = pipe.drain()
    torch.cuda.synchronize()
    assert full.shape == ref.shape and float((full - ref).abs().max()) <= 1e-11


@pytest.mark.parametrize("kernel", ["quad", "lane", "single"])
def test_minpack_gradient_test_ends_compromise_solves(golden, kernel):
    """okx_solve_opts.grad_tol < 0 is MINPACK's gtol (lmder's gnorm = max_j |(J^T r)_j| / (|J_j| |r|), what the reference's
    SolverConfig.gtol means, solver.py:158-169): beyond the reach, where the residual does not vanish, a loosened value ends
    the solve as soon as the residual is that orthogonal to the Jacobian's columns - earlier than the cost test - and the
    returned point really meets it; > 0 stays the absolute form; inside the reach neither fires."""
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c1_dw_corner")
    program = program.with_line_mode("pinned")
    dp = DeviceProgram(program, "cuda:0")
    base = arrays["targets_abs"][len(arrays["targets_abs"]) // 2]
    bump = [k for k in range(program.n_targets) if abs(program.tgt_dir[k][2]) > 0.5][0]
    t = np.repeat(base[None], 128, axis=0)
    t[:, bump] += np.concatenate([np.linspace(200.0, 500.0, 64), -np.linspace(200.0, 500.0, 64)])   # past either end of the bump travel
    kw = dict(kernel=kernel, chain_len=1, predictor=False, max_iter=400)
    probe = dp.solve(t, **kw).info()
    t = t[probe["max_residual"] > 1e-3]                   # the compromise points: targets that cannot be met
    assert len(t) >= 32

    def gnorm(res):
        x = res.positions[:, dp.free_out_index].reshape(len(t), -1)
        r, jac = dp.eval(x, t)
        r, jac = r.cpu().numpy(), jac.cpu().numpy()
        g = np.einsum("bmn,bm->bn", jac, r)
        cols = np.linalg.norm(jac, axis=1)
        return np.max(np.abs(g) / np.maximum(cols * np.linalg.norm(r, axis=1)[:, None], 1e-300), axis=1), np.max(np.abs(g), axis=1)

    default = dp.solve(t, **kw)
    loose = dp.solve(t, grad_tol=-1e-2, **kw)
    torch.cuda.synchronize()
    di, li = default.info(), loose.info()
    assert np.all(li["flags"] & 1) and np.all(default.info()["max_residual"] > 1e-3)       # converged on the test, far from feasible
    assert li["nfev"].sum() < di["nfev"].sum()
    scaled, _ = gnorm(loose)
    assert scaled.max() <= 1e-2 * (1.0 + 1e-6)
    assert gnorm(default)[0].max() <= 1e-2                   # (the default's cost test ends at least as orthogonal)
    absolute = dp.solve(t, grad_tol=5.0, **kw)               # the absolute form: max |J^T r| <= 5
    torch.cuda.synchronize()
    assert gnorm(absolute)[1].max() <= 5.0 * (1.0 + 1e-9) and np.all(absolute.info()["flags"] & 1)
    inside = arrays["targets_abs"]
    a = dp.solve(inside, kernel=kernel, chain_len=1, predictor=False)
    b = dp.solve(inside, kernel=kernel, chain_len=1, predictor=False, grad_tol=-1e-3)
    torch.cuda.synchronize()
    assert float((a.positions - b.positions).abs().max()) <= 1e-9
