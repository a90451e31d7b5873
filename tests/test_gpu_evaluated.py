"""
GPU: the EVALUATED solve (okx_solve_evaluated_batch / okx_evaluate_batch) - tangents and the metric catalog as the
solve kernels' epilogue, ONE launch (reference core/sweep.py:217-270 solve_evaluated_sweep / evaluate_solved_sweep) -
against the reference's tangent and metric goldens, against the three-launch path (solve -> okx_tangent_batch ->
okx_corner_metrics_batch) and, for the positions, bit for bit against the plain solve.
"""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, gpu_available
from test_gpu_metrics import _roles
from test_metrics_oracle import close, derivative_plan, load_metrics_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not gpu_available():
        pytest.skip("no GPU")


def _tg(name):
    return dict(np.load(os.path.join(GOLDEN, f"tangents_{name}.npz"), allow_pickle=False))


def _evaluated_program(golden, name, line_mode="pinned"):
    from open_kinematics_amd.batch import DeviceProgram

    _, program = golden(name)
    program = program.with_line_mode(line_mode)
    mg = load_metrics_golden(name)
    roles, ridx = _roles(program, mg)
    dp = DeviceProgram(program, "cuda:0")
    assert dp.kernel == "quad", dp.kernel_note
    dp.enable_evaluation(roles)
    assert dp.evaluation & 1, dp.evaluation_note
    return dp, program, roles, ridx, mg


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "e2e_sweep"])
def test_evaluate_matches_the_reference_metrics_and_derivatives(golden, name):
    """okx_evaluate_batch at the reference's own solved states: the tolerances of tests/test_gpu_metrics.py."""
    dp, program, roles, ridx, mg = _evaluated_program(golden, name)
    res = dp.evaluate(mg["pos"], tangents=True)
    torch.cuda.synchronize()
    assert np.all(res.tangent_info()["flags"] == 1)
    values = res.metrics.cpu().numpy()
    assert np.max(np.abs(values[:, :8] - mg["values"][:, :8])) <= 1e-9
    assert close(values, mg["values"], 1e-9)  # NaN exactly where the reference reports None
    deriv = res.derivatives.cpu().numpy()
    tan = res.tangents.cpu().numpy()
    plan = derivative_plan(program, mg["deriv_names"])
    assert len(plan) >= 8
    for j, (what, t) in plan.items():
        got = tan[:, t, ridx["wheel_center"], what[1]] if isinstance(what, tuple) else deriv[:, t, what]
        ref = mg["deriv"][:, j]
        assert np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-7, mg["deriv_names"][j]
    # the driver rates are the tangents' entries
    rates = res.wheel_center_rates.cpu().numpy()
    assert np.array_equal(rates, tan[:, :, ridx["wheel_center"], :])


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid"])
def test_fused_tangents_match_the_reference(golden, name):
    """Tangents of the evaluation epilogue at the reference's solved states: <= 1e-9 (tests/golden/tangents_*.npz)."""
    dp, program, roles, ridx, mg = _evaluated_program(golden, name)
    tg = _tg(name)
    res = dp.evaluate(tg["pos"], tangents=True)
    torch.cuda.synchronize()
    assert np.max(np.abs(res.tangents.cpu().numpy() - tg["vel"])) <= 1e-9
    info = res.tangent_info()
    assert np.all(info["flags"] == 1) and np.all(info["min_pivot"] > 0)


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid"])
@pytest.mark.parametrize("shape", ["cold", "chained", "ragged", "lane", "lane_ragged"])
def test_evaluated_solve_is_the_solve_plus_the_separate_launches(golden, name, shape):
    """One launch against three: positions and info records bit for bit, tangents / metrics / derivatives to rounding."""
    from open_kinematics_amd.metrics import corner_state_metrics

    dp, program, roles, ridx, mg = _evaluated_program(golden, name)
    arrays, _ = golden(name)
    t = arrays["targets_abs"]
    kw = dict(kernel="quad")
    if shape == "chained":
        kw = dict(kernel="quad", chain_len=7)
    if shape == "ragged":
        t = t[:37]
    if shape.startswith("lane"):  # one lane per problem: the catalog on duals with all T directions at once
        assert dp.evaluation & 2, dp.evaluation_note
        kw = dict(kernel="lane", chain_len=1)
        t = np.concatenate([t, t[::-1], t])[: 200 if shape == "lane" else 131]
    plain = dp.solve(t, **kw)
    fused = dp.solve_evaluated(t, tangents=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions)
    assert torch.equal(plain.info_raw, fused.info_raw)
    info = fused.info()
    assert np.all((info["flags"] & 7) == 1)
    tan, tinfo = dp.tangents(plain.positions)
    sep = corner_state_metrics(roles, plain.positions, tan)
    torch.cuda.synchronize()
    assert np.max(np.abs(fused.tangents.cpu().numpy() - tan.cpu().numpy())) <= 1e-9
    assert close(fused.metrics.cpu().numpy(), sep.values.cpu().numpy(), 1e-9)
    got, ref = fused.derivatives.cpu().numpy(), sep.derivatives.cpu().numpy()
    both = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), both)
    assert np.max(np.abs(got[both] - ref[both]) / np.maximum(1.0, np.abs(ref[both]))) <= 1e-7
    ti = dp.tangent_info(tinfo)
    fi = fused.tangent_info()
    assert np.array_equal(fi["flags"], ti["flags"])
    assert np.allclose(fi["min_pivot"], ti["min_pivot"], rtol=1e-9) and np.allclose(fi["max_pivot"], ti["max_pivot"], rtol=1e-9)
    # metrics only: nothing of the positions is written
    lean = dp.solve_evaluated(t, output="none", **kw)
    torch.cuda.synchronize()
    assert lean.positions is None and lean.tangents is None
    assert torch.equal(torch.nan_to_num(lean.eval), torch.nan_to_num(fused.eval))


@pytest.mark.parametrize("kernel", ["quad", "lane"])
def test_evaluated_ensemble_uses_each_geometrys_own_design_references(golden, kernel):
    """Geometry tables: wheel travel is measured from every geometry's own design state."""
    from open_kinematics_amd.metrics import METRIC_NAMES, corner_state_metrics

    dp, program, roles, ridx, mg = _evaluated_program(golden, "c1_dw_corner")
    rng = np.random.default_rng(3)
    g, s = (5, 16) if kernel == "quad" else (3, 70)
    hard = np.repeat(program.design_pos[None], g, axis=0) + rng.normal(0.0, 0.5, size=(g, program.n_points, 3))
    gpos, grow = dp.rebind(hard)
    rel = np.stack([np.zeros(s), np.linspace(-30.0, 40.0, s)], axis=1)
    targets = dp.ensemble_targets(gpos, rel)
    kw = dict(geom_pos=gpos, geom_row_param=grow, steps_per_geometry=s, kernel=kernel, chain_len=1)
    plain = dp.solve(targets, **kw)
    fused = dp.solve_evaluated(targets, tangents=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions)
    tan, _ = dp.tangents(plain.positions, geom_pos=gpos, geom_row_param=grow, steps_per_geometry=s)
    assert np.max(np.abs(fused.tangents.cpu().numpy() - tan.cpu().numpy())) <= 1e-9
    travel = fused.metrics[:, METRIC_NAMES.index("wheel_travel")].cpu().numpy().reshape(g, s)
    wc = program.out_point[ridx["wheel_center"]]
    z0 = gpos[:, wc, 2].cpu().numpy()
    z = plain.positions[:, ridx["wheel_center"], 2].cpu().numpy().reshape(g, s)
    assert np.max(np.abs(travel - (z - z0[:, None]))) <= 1e-12
    assert np.max(np.abs(travel - rel[None, :, 1])) <= 1e-8  # the bump target is the wheel centre's rise
    # everything else against the separate metric kernel on one geometry's block
    sep = corner_state_metrics(roles, plain.positions[:s], tan[:s])
    cols = [k for k, n in enumerate(METRIC_NAMES) if n != "wheel_travel"]
    assert close(fused.metrics[:s][:, cols].cpu().numpy(), sep.values[:, cols].cpu().numpy(), 1e-9)
    # okx_evaluate_batch with the same tables gives the same numbers as the solve's epilogue
    again = dp.evaluate(plain.positions, tangents=True, geom_pos=gpos, geom_row_param=grow, steps_per_geometry=s)
    torch.cuda.synchronize()
    assert close(again.eval.cpu().numpy(), fused.eval.cpu().numpy(), 1e-9)


def test_evaluated_entry_points_fail_loudly(golden):
    from open_kinematics_amd.batch import DeviceProgram

    arrays, program = golden("c1_dw_corner")
    dp = DeviceProgram(program.with_line_mode("pinned"), "cuda:0")
    with pytest.raises(RuntimeError, match="metric roles"):
        dp.solve_evaluated(arrays["targets_abs"])
    _, axle = golden("c3_axle_grid")
    mg = load_metrics_golden("c1_dw_corner")
    roles, _ = _roles(program.with_line_mode("pinned"), mg)
    axle_dp = DeviceProgram(axle.with_line_mode("pinned"), "cuda:0")
    with pytest.raises(ValueError, match="single-mode quad kernel"):
        axle_dp.enable_evaluation(roles)


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid", "t_corner_rocker"])
def test_solve_evaluated_sweep_is_solve_sweep_plus_compute_sweep_metrics(golden, name):
    """The drop-in (core/sweep.py:248-270): same states and statistics as solve_sweep, the rows of compute_sweep_metrics
    on those states - from ONE launch for a corner (a rocker corner's rotation metrics take a second, on the tangents)."""
    import yaml

    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.sweep import EvaluatedSweep, compute_sweep_metrics, evaluate_solved_sweep, solve_evaluated_sweep, solve_sweep

    arrays, _ = golden(name)
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    states, stats = solve_sweep(sus, sweep)
    ref = compute_sweep_metrics(sus, sweep, states)
    ev = solve_evaluated_sweep(sus, sweep)
    assert isinstance(ev, EvaluatedSweep) and len(ev.states) == len(states) == len(ev.solver_stats) == len(ev.metrics.rows)
    assert ev.diagnostics == [] and ev.metrics.derivative_error is None
    for a, b in zip(states, ev.states):
        for key, p in a.positions.items():
            assert np.array_equal(np.asarray(p.data), np.asarray(b.positions[key].data)), key
    assert [(s.converged, s.nfev) for s in stats] == [(s.converged, s.nfev) for s in ev.solver_stats]
    assert list(ev.metrics.rows[0]) == list(ref.rows[0])
    for got, want in zip(ev.metrics.rows, ref.rows):
        for key, value in want.items():
            if value is None:
                assert got[key] is None, key
            else:
                assert abs(got[key] - value) <= 1e-7 * max(1.0, abs(value)), key
    again = evaluate_solved_sweep(sus, sweep, states, stats)
    assert [list(r.items()) for r in again.metrics.rows] == [list(r.items()) for r in ref.rows]
    with pytest.raises(ValueError, match="counts must match"):
        evaluate_solved_sweep(sus, sweep, states, stats[:-1])


@pytest.mark.parametrize("name", ["c1_dw_corner", "c4_macpherson_grid"])
def test_evaluate_lane_form_matches_the_quad_form_and_the_reference(golden, name, monkeypatch):
    """okx_evaluate_batch's lane form (one lane per state; batches that fill the chip take it by themselves) against the
    quad form and the reference's goldens: ragged batch, tangents on request, per-geometry tables."""
    dp, program, roles, ridx, mg = _evaluated_program(golden, name)
    assert dp.evaluation & 2, dp.evaluation_note
    pos = np.concatenate([mg["pos"], mg["pos"][::-1], mg["pos"]])[:131]
    want = np.concatenate([mg["values"], mg["values"][::-1], mg["values"]])[:131]
    monkeypatch.setenv("OKX_DEV", "evaluate_quad")
    quad = dp.evaluate(pos, tangents=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("OKX_DEV", "evaluate_lane")
    lane = dp.evaluate(pos, tangents=True)
    lean = dp.evaluate(pos)
    torch.cuda.synchronize()
    assert close(lane.metrics.cpu().numpy(), want, 1e-9)
    assert close(lane.eval.cpu().numpy(), quad.eval.cpu().numpy(), 1e-9)
    assert np.max(np.abs(lane.tangents.cpu().numpy() - quad.tangents.cpu().numpy())) <= 1e-9
    assert torch.equal(torch.nan_to_num(lean.eval), torch.nan_to_num(lane.eval)) and lean.tangents is None
    assert np.all(lane.tangent_info()["flags"] == 1)
    # an ensemble: every geometry's own design references, wave units of one geometry each (70 steps: a ragged second unit)
    rng = np.random.default_rng(5)
    g, s = 3, 70
    hard = np.repeat(program.design_pos[None], g, axis=0) + rng.normal(0.0, 0.5, size=(g, program.n_points, 3))
    gpos, grow = dp.rebind(hard)
    rel = np.zeros((s, program.n_targets))
    rel[:, -1] = np.linspace(-20.0, 25.0, s)
    kw = dict(geom_pos=gpos, geom_row_param=grow, steps_per_geometry=s)
    solved = dp.solve(dp.ensemble_targets(gpos, rel), chain_len=1, **kw)
    lane_g = dp.evaluate(solved.positions, tangents=True, **kw)
    monkeypatch.setenv("OKX_DEV", "evaluate_quad")
    quad_g = dp.evaluate(solved.positions, tangents=True, **kw)
    torch.cuda.synchronize()
    assert close(lane_g.eval.cpu().numpy(), quad_g.eval.cpu().numpy(), 1e-9)
    assert np.max(np.abs(lane_g.tangents.cpu().numpy() - quad_g.tangents.cpu().numpy())) <= 1e-9


def test_evaluated_ensemble_gathers_metric_columns(golden, monkeypatch):
    """dist.ShardedEnsemble(metric_columns=...): every rank evaluates its shard (one launch per chunk, no positions), the
    chosen columns land in the gathered table.  One rank, and rank 0 of a world of 4 alone (exchange stubbed): its own rows
    equal the unsharded evaluated solve's, bit for bit, whatever the chunk count."""
    import open_kinematics_amd.dist as okd
    from open_kinematics_amd.metrics import METRIC_NAMES

    dp, program, roles, ridx, mg = _evaluated_program(golden, "c1_dw_corner")
    rng = np.random.default_rng(11)
    g, s = 64, 64
    table = np.repeat(program.design_pos[None], g, axis=0) + rng.normal(0.0, 0.5, size=(g, program.n_points, 3))
    table = torch.as_tensor(table, device="cuda:0")
    rel = np.stack([np.zeros(s), np.linspace(-30.0, 40.0, s)], axis=1)
    gpos, grow = dp.rebind(table)
    plan = dp.plan_launch(g * s, steps_per_geometry=s, geometry_tables=True, evaluated=True, chain_len=1, predictor=False)
    whole = dp.solve_evaluated(dp.ensemble_targets(gpos, rel), geom_pos=gpos, geom_row_param=grow, steps_per_geometry=s,
                               output="none", kernel=plan[0], chain_len=plan[1], predictor=False)
    torch.cuda.synchronize()
    cols = [("camber", None), ("camber", 1), ("roadwheel_angle", 1), (21, 1)]
    flat = whole.eval.reshape(g * s, -1)
    want = flat[:, [METRIC_NAMES.index("camber"), 48 + METRIC_NAMES.index("camber"), 48 + METRIC_NAMES.index("roadwheel_angle"), 48 + 21]]
    alone = okd.ShardedEnsemble(dp, table, rel, s, metric_columns=cols, chain_len=1, predictor=False)
    got = alone.step()
    torch.cuda.synchronize()
    assert got.shape == (g * s, 4) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
    assert torch.equal(torch.nan_to_num(alone.eval_local), torch.nan_to_num(whole.eval)) and alone.positions is None and alone.free_full is None
    assert torch.equal(alone.status_full, whole.info_raw[:, 32])

    class RankZero(okd.ShardedEnsemble):
        def _exchange_chunk(self, k):
            return []

    monkeypatch.setattr(okd, "_world", lambda group: (4, 0))
    pipe = RankZero(dp, table, rel, s, chunks=3, metric_columns=cols, chain_len=1, predictor=False)
    lo, hi = pipe.geometry_range
    own = slice(lo * s, hi * s)
    for _ in range(2):
        pipe.metric_full.zero_()
        got = pipe.step()
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(got[own]), torch.nan_to_num(want[own]))
        assert torch.equal(torch.nan_to_num(pipe.eval_local), torch.nan_to_num(whole.eval[own]))
    assert pipe.exchange_bytes_per_rank == (hi - lo) * s * (4 * 8 + 1)
    with pytest.raises(ValueError, match="no evaluation entry"):
        okd.ShardedEnsemble(dp, table, rel, s, metric_columns=[("camber", 5)])


@pytest.mark.parametrize("kernel", ["quad", "lane"])
def test_evaluated_solve_beyond_the_reach_flags_what_the_separate_launches_flag(golden, kernel):
    """Targets walked past lock-out (the reference's "did not reach an acceptable residual"): the solve's flags and
    positions stay those of the plain solve, and the tangent solve's health - ok / rank-deficient, NaN rows where the
    factorisation fails - is what okx_tangent_batch reports at the same states."""
    dp, program, roles, ridx, mg = _evaluated_program(golden, "c1_dw_corner")
    arrays, _ = golden("c1_dw_corner")
    base = arrays["targets_abs"][len(arrays["targets_abs"]) // 2]
    t = np.repeat(base[None], 192, axis=0)
    t[:, -1] += np.linspace(-400.0, 400.0, 192)      # far past either end of the bump travel
    kw = dict(kernel=kernel, chain_len=1)
    plain = dp.solve(t, **kw)
    fused = dp.solve_evaluated(t, tangents=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(plain.positions, fused.positions) and torch.equal(plain.info_raw, fused.info_raw)
    info = fused.info()
    assert np.any((info["flags"] & 7) != 1) and np.any((info["flags"] & 7) == 1)    # both kinds of state are in the batch
    tan, tinfo = dp.tangents(plain.positions)
    torch.cuda.synchronize()
    ti, fi = dp.tangent_info(tinfo), fused.tangent_info()
    agree = ti["flags"] == fi["flags"]
    # (a pivot test at the rounding floor may fall either way between two orders of summation: allow it on states the solve flagged)
    assert np.all(agree | ((info["flags"] & 7) != 1)), np.flatnonzero(~agree)
    good = (ti["flags"] == 1) & (fi["flags"] == 1) & ((info["flags"] & 7) == 1)
    assert good.sum() >= 20
    a, b = fused.tangents.cpu().numpy()[good], tan.cpu().numpy()[good]
    assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= 1e-7
    failed = (fi["flags"] & 1) == 0
    if failed.any():  # a failed factorisation reads NaN in every derivative, never a number
        assert np.all(np.isnan(fused.derivatives.cpu().numpy()[failed]))


@pytest.mark.parametrize("name", ["c1_dw_corner", "t_axle_dw"])
def test_solve_evaluated_sweep_without_evaluated_kernels_falls_back_to_solve_then_evaluate(golden, name, monkeypatch):
    """A program whose evaluated kernels cannot be had (no single-mode / pair-mode quad kernel, more role points than fit, a
    compile failure: enable_evaluation raises) is solved first and evaluated after by the separate launches - the documented
    fallback, with the rows of the fused path.  (Round 5 compared two bound methods with `is` there and died with an
    AttributeError instead.)"""
    import yaml

    from open_kinematics_amd import batch
    from open_kinematics_amd.input import build_suspension, build_sweep
    from open_kinematics_amd.sweep import solve_evaluated_sweep

    arrays, _ = golden(name)
    sus = build_suspension(yaml.safe_load(str(arrays["geometry_yaml"])))
    sweep = build_sweep(yaml.safe_load(str(arrays["sweep_yaml"])), sus)
    fused = solve_evaluated_sweep(sus, sweep)

    def refuse(self, roles):
        raise ValueError("no evaluated kernels for this program: (test)")

    monkeypatch.setattr(batch.DeviceProgram, "enable_evaluation", refuse)
    plain = solve_evaluated_sweep(sus, sweep)
    assert len(plain.states) == len(fused.states) == len(plain.metrics.rows) and plain.metrics.derivative_error is None
    for a, b in zip(plain.states, fused.states):
        for key, p in a.positions.items():
            assert np.array_equal(np.asarray(p.data), np.asarray(b.positions[key].data)), key

    def flat(row):
        return row.flat_row() if hasattr(row, "flat_row") else row

    for got, want in zip(plain.metrics.rows, fused.metrics.rows):
        got, want = flat(got), flat(want)
        assert list(got) == list(want)
        for key, value in want.items():
            assert (got[key] is None and value is None) or abs(got[key] - value) <= 1e-7 * max(1.0, abs(value)), key
