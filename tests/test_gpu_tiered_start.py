"""
Tiered start (okx.h, okx_program_ready): a program whose generated kernels are not in the kernel cache is usable at once on
the interpreter kernels while a host thread compiles; when the job is done the program switches over, and the answers
before and after the switch agree to 1e-9 mm.  The drop-in's first solve_sweep of an uncached axle returns in well under a
second instead of waiting for the compiler (10 ... 80 s per module).
"""

import os
import time

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_uncached_program_solves_at_once_and_switches_over(golden, monkeypatch, tmp_path):
    from open_kinematics_amd.batch import DeviceProgram

    monkeypatch.setenv("OKX_KERNEL_CACHE", str(tmp_path))       # an empty kernel cache: nothing of this program is compiled
    monkeypatch.setenv("OKX_DEV", "no_lane")                    # (one module to compile is enough for the test)
    arrays, program = golden("c4_macpherson_grid")
    program = program.with_line_mode("pinned")
    targets = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    t0 = time.perf_counter()
    dp = DeviceProgram(program, "cuda:0", wait_for_kernels=False)
    created = time.perf_counter() - t0
    assert created < 2.0, f"okx_program_create took {created:.1f} s"
    assert not dp.ready and dp.kernel == "wave" and "compiled" in dp.kernel_note
    before = dp.solve(targets, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    info_before = before.info()
    assert before.accepted(info_before).all()
    assert dp.kernel == "wave"                                   # the compiler is not done within a solve's time
    dp.wait_ready()                                              # ... now it is: the program has switched over
    assert dp.ready and dp.kernel == "quad", dp.kernel_note
    assert any(name.endswith(".okxc") for name in os.listdir(tmp_path))
    after = dp.solve(targets, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    assert after.accepted(after.info()).all()
    assert float((before.positions - after.positions).abs().max()) <= 1e-9
    # a second program of the same source finds the cache: generated kernels from the first call on
    again = DeviceProgram(program, "cuda:0", wait_for_kernels=False)
    assert again.ready and again.kernel == "quad"
    again.close()
    dp.close()


def test_first_solve_sweep_of_an_uncached_axle_does_not_wait_for_the_compiler(monkeypatch, tmp_path):
    from open_kinematics_amd import solver
    from open_kinematics_amd.input import load_geometry, load_sweep
    from open_kinematics_amd.sweep import solve_sweep

    geom = os.path.join(GOLDEN, "geometry")
    axle = load_geometry(os.path.join(geom, "axle_geometry.yaml"))
    sweep = load_sweep(os.path.join(geom, "axle_sweep.yaml"), axle)
    solver.clear_program_cache()
    warm = solve_sweep(axle, sweep)                              # kernels from the in-tree cache: the reference answer
    solver.clear_program_cache()
    monkeypatch.setenv("OKX_KERNEL_CACHE", str(tmp_path))
    t0 = time.perf_counter()
    states, infos = solve_sweep(axle, sweep)
    elapsed = time.perf_counter() - t0
    assert elapsed < 0.5, f"first solve_sweep of an uncached axle took {elapsed:.2f} s"
    assert all(i.converged for i in infos)
    dp = next(iter(solver._PROGRAM_CACHE.values()))
    assert dp.kernel == "wave"                                   # served by the interpreter kernels
    out = axle.output_points()
    a = np.array([[s.positions[k].data for k in out] for s in states])
    b = np.array([[s.positions[k].data for k in out] for s in warm[0]])
    assert np.abs(a - b).max() <= 1e-9
    dp.wait_ready()                                              # (the module lands in tmp_path; without this the job would be
    solver.clear_program_cache()                                 #  joined at interpreter exit: destroying a program never waits)
    del dp


def test_switch_over_under_concurrent_launches_and_without_a_writable_cache(golden, monkeypatch, tmp_path):
    """The compile job hands its code objects over in memory (a read-only cache directory costs nothing but the caching),
    and the switch-over is safe against launches of the same program from other threads."""
    import stat
    import threading

    from open_kinematics_amd.batch import DeviceProgram

    locked = tmp_path / "readonly"
    locked.mkdir()
    locked.chmod(stat.S_IRUSR | stat.S_IXUSR)
    if os.access(locked, os.W_OK):
        pytest.skip("this user writes to read-only directories (root)")
    monkeypatch.setenv("OKX_KERNEL_CACHE", str(locked))
    monkeypatch.setenv("OKX_DEV", "no_lane")
    arrays, program = golden("c4_macpherson_grid")
    program = program.with_line_mode("pinned")
    host_targets = arrays["targets_abs"]
    dp = DeviceProgram(program, "cuda:0", wait_for_kernels=False)
    assert not dp.ready and dp.kernel == "wave"
    reference = dp.solve(torch.as_tensor(host_targets, device="cuda:0"), chain_len=1, predictor=False).positions.clone()
    torch.cuda.synchronize()
    stop, errors, seen = threading.Event(), [], set()

    def worker():
        try:
            stream = torch.cuda.Stream(device="cuda:0")
            targets = torch.as_tensor(host_targets, device="cuda:0")
            with torch.cuda.stream(stream):
                while not stop.is_set():
                    res = dp.solve(targets, chain_len=1, predictor=False)
                    stream.synchronize()
                    seen.add(dp.kernel)
                    if not res.accepted(res.info()).all() or float((res.positions - reference).abs().max()) > 1e-9:
                        errors.append("a solve during the switch-over gave other answers")
                        return
        except Exception as error:  # noqa: BLE001
            errors.append(f"{type(error).__name__}: {error}")

    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    deadline = time.perf_counter() + 240.0
    while not dp.ready and time.perf_counter() < deadline:       # (`ready` itself switches over once the job is done)
        time.sleep(0.05)
    time.sleep(0.3)                                              # ... and some launches on the generated kernels
    stop.set()
    for t in threads:
        t.join()
    assert not errors, errors
    assert dp.ready and dp.kernel == "quad", dp.kernel_note      # switched over from memory: the cache could not be written
    assert seen == {"wave", "quad"}
    assert os.listdir(locked) == []
    locked.chmod(stat.S_IRWXU)
    dp.close()


def test_quad_kernels_serve_while_the_lane_module_still_compiles(golden, monkeypatch, tmp_path):
    """Two stages: the quad module is switched over to as soon as IT is compiled; the lane module (its emission variants
    take the compiler another while) follows when the whole job is done.  Same answers in all three phases."""
    from open_kinematics_amd.batch import DeviceProgram

    monkeypatch.setenv("OKX_KERNEL_CACHE", str(tmp_path))
    monkeypatch.delenv("OKX_DEV", raising=False)
    arrays, program = golden("c4_macpherson_grid")
    program = program.with_line_mode("pinned")
    targets = torch.as_tensor(arrays["targets_abs"], device="cuda:0")
    dp = DeviceProgram(program, "cuda:0", wait_for_kernels=False)
    assert not dp.ready and dp.kernel == "wave"
    first = dp.solve(targets, chain_len=1, predictor=False).positions.clone()
    torch.cuda.synchronize()
    quad_while_pending = False
    deadline = time.perf_counter() + 300.0
    while time.perf_counter() < deadline:
        res = dp.solve(targets, chain_len=1, predictor=False)    # (every launch looks for a finished stage)
        torch.cuda.synchronize()
        assert float((res.positions - first).abs().max()) <= 1e-9
        kernel, pending = dp.kernel, not dp.ready
        if kernel == "quad" and pending:
            quad_while_pending = True
        if not pending:
            break
        time.sleep(0.25)
    assert dp.ready and dp.kernel == "quad", dp.kernel_note
    assert quad_while_pending, "the quad kernels were only attached together with the lane module"
    assert dp.lane_bodies & 1, dp.lane_note                      # ... and the lane kernels did arrive
    big = targets.repeat(80, 1)[:20000].contiguous()             # a batch the lane kernel takes
    res = dp.solve(big, chain_len=1, predictor=False)
    torch.cuda.synchronize()
    assert res.accepted(res.info()).all()
    dp.close()
