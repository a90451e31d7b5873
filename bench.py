#!/usr/bin/env python3
"""
Headline benchmark: constraint solves/sec on the double-wishbone bump sweep (BASELINE.json).

One "step" = one pass of the hot path over one batch: every rank solves a 16384-step fp64 bump sweep
(BASELINE config 2; inputs already resident in HBM) with ONE launch of the solve kernel (the
runtime-specialised quad kernel `okx_quad_solve_u`).  Every one of the 16384 solves is an
independent COLD START FROM THE DESIGN STATE — SURVEY.md section 8(d)'s start rule, the reference's own
(`core/solver.py:710`) — and that launch is what `value`, `roofline` and `compute` describe.

The same JSON line also carries, each measured in this run:
  own_first_pass the same launch with shared_first_step=0 (every problem evaluates the design state itself)
  pipelined      the same cold sweeps round-robin over three streams (throughput of independent sweeps)
  with_model     the same sweep with chain heads started from the fitted Chebyshev model (okx_program_fit_predictor),
                 WITH the cost of the fit (fit_ms) and of a one-shot sweep (first_sweep_ms) on the record
  e2e            host buffers -> H2D targets -> solve -> D2H positions + info (pinned memory), per leg
  cpu_baseline   the CPU oracle (reference-shaped MINPACK sweeps) on this box's host cores: one core timed,
                 then every core the process may run on
  other_configs  BASELINE configs 3, 4, 5 at full size: cold independent solves and the product's chained mode
  downstream     tangents and the metric catalog on the solved states
  dropin         wall-clock of the drop-in `solve_sweep` on the reference's own benchmark workloads

With N > 1 ranks the global sweep (N x 16384 steps) is sharded by index and the solved free coordinates are
all-gathered over RCCL inside the timed step (`value` is exchange-inclusive; `solve_only` is reported beside it).
`--config c5` runs BASELINE config 5 instead (4096 perturbed geometries x 256 steps, geometry-major shards).

  python bench.py --gpus 1 --steps 200 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
         --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

STEPS_PER_RANK = 16384
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # vendor fp64 vector peak, secondary ceiling
PROFILE_ROUND = "r06"
INFO_FIELDS = np.dtype([("max_residual", "<f8"), ("cost", "<f8"), ("last_step", "<f8"), ("iterations", "<i4"),
                        ("nfev", "<i4"), ("flags", "<i4"), ("reserved", "<i4")])


def algorithmic_bytes_per_solve(program, steps_per_geometry: int = 0, authored_points: int = 10, params: int = 17) -> float:
    """8*T targets in + 24*P_out positions out + 16 B info (we write 40), plus for ensembles the
    per-geometry inputs (authored points x 24 B + row parameters x 8 B) amortised over that geometry's steps."""
    base = 8 * program.n_targets + 24 * program.n_out + 16
    if steps_per_geometry > 0:
        base += (24 * authored_points + 8 * params) / steps_per_geometry
    return float(base)


def estimated_flops_per_evaluation(program, stats) -> float:
    """ANALYTIC fp64 flop count of one LM iteration: rows + J^T J + LDL^T + substitutions.
    A model, not a counter reading; the PMC instruction counts are in profiles/ (SQ_INSTS_VALU)."""
    n, m = program.n_vars, program.n_residuals
    rows = 60.0 * m
    normal = 2.0 * 9.0 * stats["contrib"] + 2.0 * 3.0 * stats["contrib"]
    chol = n ** 3 / 3.0 + 2.0 * n * n
    return rows + normal + chol


def plan_stats(program) -> dict:
    import ctypes as C

    from open_kinematics_amd import _lib
    from open_kinematics_amd._abi import HostProgram

    raw = (C.c_int32 * 8)()
    _lib.load().okx_debug_plan_stats(HostProgram(program).byref(), raw)
    return dict(zip(["n", "m", "pairs", "contrib", "active", "js_stride", "lda", "lds_bytes"], list(raw)))


def committed_fp64(tag: str):
    """fp64 instruction counts of the solve kernel from the committed PMC pass of this same command (SQ_INSTS_VALU_ADD/MUL/
    FMA_F64, wave-level, per-dispatch medians): the counter-based companion of the analytic flop model."""
    path = os.path.join(REPO, "profiles", PROFILE_ROUND, f"{tag}_pmc_traffic.json")
    try:
        with open(path, "r", encoding="utf-8") as fh:
            return json.load(fh).get("fp64_instructions_per_launch"), os.path.relpath(path, REPO)
    except (OSError, ValueError):
        return None, None


def committed_traffic(tag: str) -> tuple:
    """
    HBM bytes per launch of the solve kernel from the COMMITTED rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE collected in separate runs; profiles/<round>/<tag>_pmc_traffic.json holds the command,
    units and corrections).  bench.py cannot run the profiler on itself, so the figure is read back from that
    summary and labelled as not measured in this run; a configuration without a committed profile reports null.
    """
    path = os.path.join(REPO, "profiles", PROFILE_ROUND, f"{tag}_pmc_traffic.json")
    try:
        with open(path, "r", encoding="utf-8") as fh:
            summary = json.load(fh)
        return float(summary["traffic_bytes_per_launch"]), os.path.relpath(path, REPO)
    except (OSError, KeyError, ValueError):
        return None, None


def time_launches(launch, steps: int, warmup: int, device) -> tuple:
    """(wall seconds per step, kernel ms per launch from ONE HIP-event pair around the K back-to-back launches on
    the launch stream — torch's current stream is the stream the C-ABI call launches on)."""
    for _ in range(warmup):
        launch()
    torch.cuda.synchronize(device)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record()
    for _ in range(steps):
        launch()
    end.record()
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / steps
    return wall, start.elapsed_time(end) / steps


def info_summary(info_tensor) -> tuple:
    host = info_tensor.cpu().numpy().view(INFO_FIELDS).reshape(-1)
    return float(host["nfev"].mean()), bool(np.all((host["flags"] & 7) == 1))


# --------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on this box's host cores, in ONE child process that never touches the GPU
# --------------------------------------------------------------------------------------------------

_CPU_CODE = """
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, {repo!r})
from open_kinematics_amd.workloads import bump_sweep_problem
from oracle.oracle import Oracle
n, cores = {n}, {cores}
program, targets = bump_sweep_problem(n, line_mode="softnorm")
orc = Oracle(program)
orc.sweep(targets[:64])  # warm the library / caches
def chain(lo, hi):
    t0 = time.perf_counter()
    res = orc.sweep(targets[lo:hi])   # one C call (ctypes releases the GIL): a sequential warm-started chain
    return time.perf_counter() - t0, int(res.first_failed_step), float(res.info["nfev"].sum())
one = chain(0, n)
bounds = [(n * k // cores, n * (k + 1) // cores) for k in range(cores)]
best = None
with ThreadPoolExecutor(cores) as pool:
    for _ in range(3):
        t0 = time.perf_counter()
        parts = list(pool.map(lambda b: chain(*b), bounds))
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, parts)
print(json.dumps({{"one": one, "all_wall": best[0], "all_parts": best[1]}}))
"""


from open_kinematics_amd.hostcpu import fit_host_threads, host_cores  # noqa: E402  (re-exported: tests/test_host_api.py)


def cpu_baseline(n_steps: int) -> dict:
    """
    Oracle = reference-shaped CPU path (sequential warm-started MINPACK LM, reference default tolerances) timed on the
    GPU box's host cores: (i) ONE core walking the whole sweep as one chain, exactly the
    reference's shape; (ii) every core this process may run on (host_cores: affinity mask / cgroup quota, no other cap), the sweep cut into
    one contiguous warm-started chain per core, best of three.  One child interpreter with HIP_VISIBLE_DEVICES=""
    (the oracle is plain C behind ctypes, the chains run as threads of that child).
    """
    cores, how = host_cores()
    cores = max(1, min(cores, n_steps // 16))
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", _CPU_CODE.format(repo=REPO, n=n_steps, cores=cores)], env=env,
                         capture_output=True, text=True, timeout=600, check=True).stdout
    res = json.loads(out.strip().splitlines()[-1])
    one_s, one_fail, one_nfev = res["one"]
    if one_fail != -1 or any(p[1] != -1 for p in res["all_parts"]):
        raise RuntimeError("cpu baseline: oracle sweep failed")
    nfev_all = sum(p[2] for p in res["all_parts"]) / n_steps
    return {
        "value": n_steps / res["all_wall"],
        "unit": "constraint solves/s",
        "cores": cores,
        "kind": "port",
        "sample": f"full {n_steps}-step bump sweep cut into {cores} contiguous warm-started chains, one thread per host "
                  f"core ({how}), MINPACK LM ftol=1e-5 xtol=gtol=1e-9 (reference defaults), wall "
                  f"{res['all_wall']:.3f} s (best of 3), mean nfev {nfev_all:.1f}",
        "one_core": {
            "value": n_steps / one_s,
            "cores": 1,
            "sample": f"the same {n_steps}-step sweep as ONE sequential warm-started chain on one core (the reference's "
                      f"shape, solver.py:716,774): {one_s:.2f} s, mean nfev {one_nfev / n_steps:.1f}",
        },
    }


# --------------------------------------------------------------------------------------------------
# single-GPU extras
# --------------------------------------------------------------------------------------------------

def measure_with_model(program, targets, device, steps: int, warmup: int) -> dict:
    """The predictor path with its setup cost on the record: a fresh program, the fit (8 node solves + host fit,
    synchronous), the first sweep, then K steady-state launches."""
    from open_kinematics_amd.batch import DeviceProgram

    dp = DeviceProgram(program, device)
    n = targets.shape[0]
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    # one cold sweep first: the fresh program's code object is loaded lazily at its first launch (tens of ms, paid by any
    # first launch, model or not); what is timed below is what the fit adds on top
    dp.plan(targets, out=out, info_out=info, chain_len=-1, predictor=False)()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    fitted = dp.fit_predictor(targets)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    if not fitted:
        dp.close()
        return {"available": False}
    launch = dp.plan(targets, out=out, info_out=info, chain_len=-1, predictor=True)
    launch()
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    wall, kernel_ms = time_launches(launch, steps, warmup, device)
    nfev, ok = info_summary(info)
    dp.close()
    return {
        "available": True,
        "value": n / wall,
        "kernel_ms": kernel_ms,
        "fit_ms": (t1 - t0) * 1e3,
        "first_sweep_ms": (t2 - t0) * 1e3,
        "lm_evaluations_mean": nfev,
        "all_converged": ok,
        "note": "chain heads start at the degree-7 Chebyshev model of the solution over THIS sweep's target box, fitted "
                "from 8 node solves (okx_program_fit_predictor); value = steady-state launches after the fit; "
                "first_sweep_ms = fit + one sweep, i.e. what a one-shot sweep pays (compare ms_per_step of the cold launch)",
    }


def measure_one_shot(program, targets, device, steady_kernel_ms: float) -> dict:
    """What a caller who makes ONE call observes on a fresh program, and what the shared first step costs.  The table of
    the program's own geometry for the default damping is part of okx_program_create (one wavefront, synchronous), so the
    first launch is a plain solve; a launch with another lambda0 fills a new table on its stream first (head_ms)."""
    from open_kinematics_amd.batch import DeviceProgram

    n = targets.shape[0]
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    dp = DeviceProgram(program, device)   # code objects come from the in-tree cache; includes the own-geometry table
    torch.cuda.synchronize(device)
    create_ms = (time.perf_counter() - t0) * 1e3

    def timed_once(**kw):
        launch = dp.plan(targets, out=out, info_out=info, chain_len=-1, predictor=False, **kw)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for e in ev:   # first record() of an event creates it: not inside the measurement
            e.record()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        ev[0].record()
        launch()
        ev[1].record()
        torch.cuda.synchronize(device)
        return ev[0].elapsed_time(ev[1]), (time.perf_counter() - t1) * 1e3

    first_gpu_ms, first_wall_ms = timed_once()                      # the fresh program's first launch (lazy code-object load included)
    second_gpu_ms, _ = timed_once()                                 # ... and its second
    other_gpu_ms, _ = timed_once(lambda0=1.0000001e-6)              # a damping without a table yet: head kernel + solve
    nfev, ok = info_summary(info)
    dp.close()
    return {"program_create_ms": create_ms, "first_launch_gpu_ms": first_gpu_ms, "first_launch_wall_ms": first_wall_ms,
            "second_launch_gpu_ms": second_gpu_ms, "head_ms": max(0.0, other_gpu_ms - second_gpu_ms),
            "launch_with_new_lambda0_gpu_ms": other_gpu_ms, "value_first_launch": n / (first_gpu_ms * 1e-3),
            "all_converged": ok,
            "note": "fresh DeviceProgram (kernels from the in-tree cache): create (upload, module load, own-geometry first-step "
                    "table for the default lambda0: synchronous, part of set-up), then single launches timed with one HIP-event "
                    "pair each; head_ms = what a launch pays when its lambda0 has no table yet (okx_quad_head_u on the launch "
                    "stream) over the steady launch; the headline's kernel_ms is %.4f" % steady_kernel_ms}


def measure_pipelined(dp, targets, device, steps: int, n_streams: int = 3) -> dict:
    """The same cold sweeps issued round-robin over a few streams (own output buffers): a launch no longer waits for its
    predecessor to drain, so the ~3.5 us dispatch gap and the prologue's memory round trip hide behind the previous
    sweep's tail.  Throughput of INDEPENDENT sweeps; the single-stream figure stays the headline (its kernel time is what
    `roofline` prices, and overlapped kernels have no clean per-launch duration)."""
    p = dp.program
    n = targets.shape[0]
    streams = [torch.cuda.Stream(device) for _ in range(n_streams)]
    plans = []
    for stream in streams:
        with torch.cuda.stream(stream):
            out = torch.empty((n, p.n_out, 3), dtype=torch.float64, device=device)
            info = torch.empty((n, 40), dtype=torch.uint8, device=device)
            plans.append(dp.plan(targets, out=out, info_out=info, chain_len=-1, predictor=False))
    torch.cuda.synchronize(device)
    for k in range(4 * n_streams):
        plans[k % n_streams]()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        plans[k % n_streams]()
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / steps
    return {"value": n / wall, "us_per_sweep": wall * 1e6, "streams": n_streams,
            "note": "independent cold sweeps round-robin over several streams and output buffers; not the headline"}


def measure_e2e(dp, targets_host: np.ndarray, device, steps: int, cold_kw: dict) -> dict:
    """Host buffers in, host buffers out (pinned): H2D of the targets, the solve, D2H of positions + info."""
    p = dp.program
    n = targets_host.shape[0]
    h_t = torch.as_tensor(targets_host).pin_memory()
    d_t = torch.empty_like(h_t, device=device)
    d_out = torch.empty((n, p.n_out, 3), dtype=torch.float64, device=device)
    d_info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    h_out = torch.empty((n, p.n_out, 3), dtype=torch.float64).pin_memory()
    h_info = torch.empty((n, 40), dtype=torch.uint8).pin_memory()
    launch = dp.plan(d_t, out=d_out, info_out=d_info, **cold_kw)

    def once(events=None):
        if events:
            events[0].record()
        d_t.copy_(h_t, non_blocking=True)
        if events:
            events[1].record()
        launch()
        if events:
            events[2].record()
        h_out.copy_(d_out, non_blocking=True)
        h_info.copy_(d_info, non_blocking=True)
        if events:
            events[3].record()

    for _ in range(3):
        once()
    torch.cuda.synchronize(device)
    per_sweep = np.empty(steps)
    t0 = time.perf_counter()
    for k in range(steps):
        once()
        torch.cuda.synchronize(device)  # the caller owns the host buffers again after every sweep
        t1 = time.perf_counter()
        per_sweep[k], t0 = t1 - t0, t1
    wall = float(per_sweep.mean())
    median = float(np.median(per_sweep))
    legs = np.zeros(3)
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        once(ev)
        torch.cuda.synchronize(device)
        legs += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])]
    legs /= steps
    return {
        "value": n / wall,
        "ms_per_sweep": wall * 1e3,
        # the spread of the sweeps of this one run: a stalled copy command (tens of ms, the copy engine's completion signal
        # on a shared host) inside a window of a few ms of work moves the mean by multiples and the median not at all
        "value_at_median": n / median,
        "ms_per_sweep_median": median * 1e3,
        "ms_per_sweep_max": float(per_sweep.max() * 1e3),
        "sweeps_over_3x_median": int(np.sum(per_sweep > 3.0 * median)),
        "slowest_sweep_index": int(np.argmax(per_sweep)),
        "h2d_ms": float(legs[0]),
        "kernel_ms": float(legs[1]),
        "d2h_ms": float(legs[2]),
        "bytes_h2d": int(h_t.numel() * 8),
        "bytes_d2h": int(h_out.numel() * 8 + h_info.numel()),
        "note": "pinned host buffers, one stream, synchronised after every sweep; never reported as `value`",
    }


def measure_e2e_compact(dp, targets_host: np.ndarray, device, steps: int, cold_kw: dict) -> dict:
    """Host buffers in, host buffers out, with the boundary's compact output (okx_solve_opts.output = free coordinates:
    144 B + the 40-byte info record per double-wishbone solve instead of 360 + 40 B) and three sets of buffers on three
    streams, so that the D2H of sweep k overlaps the H2D and the solve of sweep k + 1.  The host gets the free points;
    `expand` (device) or a host-side re-evaluation of the derived points rebuilds full records where they are wanted."""
    p = dp.program
    n = targets_host.shape[0]
    h_t = torch.as_tensor(targets_host).pin_memory()
    slots = []
    n_slots = 3
    for _ in range(n_slots):
        stream = torch.cuda.Stream(device)
        with torch.cuda.stream(stream):
            d_t = torch.empty_like(h_t, device=device)
            # free coordinates and info records side by side in ONE device buffer and one pinned host buffer: one D2H copy
            free_bytes = n * p.n_free * 24
            d_blob = torch.empty(free_bytes + n * 40, dtype=torch.uint8, device=device)
            d_free = d_blob[:free_bytes].view(torch.float64).view(n, p.n_free, 3)
            d_info = d_blob[free_bytes:].view(n, 40)
            launch = dp.plan(d_t, out=d_free, info_out=d_info, output="free", **cold_kw)
        h_blob = torch.empty(free_bytes + n * 40, dtype=torch.uint8).pin_memory()
        slots.append(dict(stream=stream, d_t=d_t, d_blob=d_blob, launch=launch, h_blob=h_blob,
                          h_free=h_blob[:free_bytes].view(torch.float64).view(n, p.n_free, 3), h_info=h_blob[free_bytes:].view(n, 40),
                          done=torch.cuda.Event()))

    def body(slot):
        slot["d_t"].copy_(h_t, non_blocking=True)
        slot["launch"]()
        slot["h_blob"].copy_(slot["d_blob"], non_blocking=True)

    # One HIP graph per buffer set (H2D, the solve launch through the C-ABI, two D2H copies): replaying it costs the host
    # one call instead of four stream operations, which is what bounds a ~70 us sweep from Python.  Falls back to plain
    # stream operations when the capture is refused.
    how = "hip graph replay per sweep"
    try:
        for slot in slots:
            with torch.cuda.stream(slot["stream"]):
                body(slot)   # warm: nothing of the first launch (lazy loads) inside the capture
            slot["stream"].synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=slot["stream"]):
                body(slot)
            slot["graph"] = graph
    except Exception as exc:  # pragma: no cover - depends on the runtime
        how = f"stream operations (graph capture refused: {type(exc).__name__})"
        for slot in slots:
            slot.pop("graph", None)

    def issue(slot):
        if "graph" in slot:
            slot["graph"].replay()
            slot["done"].record(slot["stream"])
        else:
            with torch.cuda.stream(slot["stream"]):
                body(slot)
                slot["done"].record()

    for k in range(2 * n_slots):
        issue(slots[k % n_slots])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        slot = slots[k % n_slots]
        slot["done"].synchronize()   # the host owns this slot's buffers again (n_slots sweeps ago)
        issue(slot)
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / steps
    ok = bool(np.all((slots[0]["h_info"].numpy().view(INFO_FIELDS).reshape(-1)["flags"] & 7) == 1))
    return {"value": n / wall, "ms_per_sweep": wall * 1e3, "bytes_h2d": int(h_t.numel() * 8),
            "bytes_d2h": int(n * (p.n_free * 24 + 40)), "all_converged": ok, "host_side": how,
            "note": "output = free coordinates, three buffer sets on three streams (D2H of one sweep under the next ones' H2D + solve), "
                    "pinned host buffers; never reported as `value`"}


def zero_copy_buffers(program, targets_host: np.ndarray, n_slots: int = 3) -> list:
    """Pinned host buffers of measure_e2e_zero_copy: (targets, free coordinates, info records) per slot."""
    n = targets_host.shape[0]
    h_t = torch.as_tensor(targets_host).pin_memory()
    return [(h_t, torch.empty((n, program.n_free, 3), dtype=torch.float64).pin_memory(),
             torch.empty((n, 40), dtype=torch.uint8).pin_memory()) for _ in range(n_slots)]


def measure_e2e_zero_copy(dp, targets_host: np.ndarray, device, steps: int, cold_kw: dict, n_slots: int = 3, buffers=None) -> dict:
    """Host buffers in, host buffers out with NO copy commands: the boundary takes plain pointers, and pinned host memory
    is device-accessible, so the solve kernel reads its targets from and stores its compact output (free coordinates +
    info records) straight into the caller's pinned buffers - the bytes cross PCIe as the kernel issues them, three
    sweeps in flight on three streams.  What bounds it is the link (3.0 MB out per 16384-step sweep), not the copy
    engine's per-command latency that bounds the copy-based pipeline above."""
    p = dp.program
    n = targets_host.shape[0]
    buffers = buffers or zero_copy_buffers(p, targets_host, n_slots)
    h_t = buffers[0][0]
    slots = []
    for _, h_free, h_info in buffers:
        stream = torch.cuda.Stream(device)
        with torch.cuda.stream(stream):
            launch = dp.plan(h_t, out=h_free, info_out=h_info, output="free", zero_copy=True, **cold_kw)
        slots.append(dict(stream=stream, launch=launch, h_free=h_free, h_info=h_info, done=torch.cuda.Event()))

    def issue(slot):
        with torch.cuda.stream(slot["stream"]):
            slot["launch"]()
            slot["done"].record()

    for k in range(2 * n_slots):
        issue(slots[k % n_slots])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        slot = slots[k % n_slots]
        slot["done"].synchronize()   # the host owns this slot's buffers again (n_slots sweeps ago)
        issue(slot)
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / steps
    ok = bool(np.all((slots[0]["h_info"].numpy().view(INFO_FIELDS).reshape(-1)["flags"] & 7) == 1))
    # the same sweep through device buffers: the host copy must hold the same bits
    ref = dp.solve(torch.as_tensor(targets_host, device=device), output="free", **cold_kw)
    torch.cuda.synchronize(device)
    same = bool(np.array_equal(slots[0]["h_free"].numpy(), ref.free.cpu().numpy()))
    bytes_out = int(n * (p.n_free * 24 + 40))
    return {"value": n / wall, "ms_per_sweep": wall * 1e3, "bytes_in_over_pcie": int(h_t.numel() * 8), "bytes_out_over_pcie": bytes_out,
            "pcie_out_gbs": bytes_out / wall / 1e9, "all_converged": ok, "same_bits_as_device_buffers": same, "streams": len(buffers),
            "note": "d_targets / d_out_pos / d_info of okx_solve_batch are the caller's pinned host buffers (device-accessible): "
                    "no H2D / D2H commands at all; never reported as `value`"}


def measure_config(name: str, make, device, steps: int, warmup: int, modes=("cold", "chained"), roles_geometry: str | None = None) -> dict:
    """One BASELINE configuration at full size: cold independent solves (section 8d's rule) and the product's chained mode.
    `roles_geometry` (a corner's geometry file): also the EVALUATED launch - solve + tangents + metric catalog with derivative
    columns as one kernel (okx_solve_evaluated_batch), against the same as three launches."""
    from open_kinematics_amd.batch import DeviceProgram

    t0 = time.perf_counter()
    made = make()
    kw, spg = {}, 0
    if len(made) == 3:  # ensemble: (program, hardpoint table, relative targets)
        program, table, rel = made
        dp = DeviceProgram(program, device)
        spg = rel.shape[0]
        table_dev = torch.as_tensor(table, device=device)
        for _ in range(2):  # the second round is the one reported: the first pays the lazy load of every kernel involved
            torch.cuda.synchronize(device)
            r0 = time.perf_counter()
            gpos, gparam = dp.rebind(table_dev)
            targets = dp.ensemble_targets(gpos, rel)
            torch.cuda.synchronize(device)
            rebind_ms = (time.perf_counter() - r0) * 1e3
        kw = dict(geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=spg)
    else:
        program, targets_host = made
        dp = DeviceProgram(program, device)
        targets = torch.as_tensor(targets_host, device=device)
        rebind_ms = None
    setup_ms = (time.perf_counter() - t0) * 1e3
    n = targets.shape[0]
    out = torch.empty((n, program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n, 40), dtype=torch.uint8, device=device)
    bytes_per = algorithmic_bytes_per_solve(program, spg)
    lane_from = dp.lane_threshold
    lane_bodies = dp.lane_bodies if lane_from > 0 else 0

    def kernel_of(cold: bool) -> str:
        if dp.kernel != "quad":
            return f"{dp.kernel} ({dp.kernel_note})"
        if lane_from > 0 and n >= lane_from and (lane_bodies & (1 if cold else 2)):
            return "lane (one lane per problem, 64 per wavefront)"
        if lane_from > 0 and n >= lane_from and not cold and (lane_bodies & 1):
            return "lane, independent solves (chain_len -1 resolves to cold starts: this program's chain body spills)"
        return "quad (four lanes per problem)" + (f"; lane kernel: {dp.lane_note}" if dp.lane_note else "")

    res = {"workload": name, "problems": n, "n_vars": program.n_vars, "n_residual_rows": program.n_residuals,
           "kernel": dp.kernel + ("" if dp.kernel == "quad" else f" ({dp.kernel_note})"),
           "algorithmic_bytes_per_solve": bytes_per, "setup_ms": setup_ms}
    if rebind_ms is not None:
        res["rebind_ms"] = rebind_ms
    for tag, chain_len in (("cold", 1), ("chained", -1)):
        if tag not in modes:
            continue
        launch = dp.plan(targets, out=out, info_out=info, chain_len=chain_len, predictor=False, **kw)
        warm_until = time.perf_counter() + 0.05  # >= 50 ms of the same launch first: the clocks ramp over tens of ms
        while time.perf_counter() < warm_until:
            launch()
            torch.cuda.synchronize(device)
        wall, kernel_ms = time_launches(launch, steps, warmup, device)
        nfev, ok = info_summary(info)
        gbs = bytes_per * n / (kernel_ms * 1e-3) / 1e9
        res[tag] = {"value": n / wall, "kernel_ms": kernel_ms, "lm_evaluations_mean": nfev, "all_converged": ok,
                    "algorithmic_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "kernel": kernel_of(chain_len == 1)}
        if not os.environ.get("OKX_BENCH_NO_QUAD_COMPARE") and dp.kernel == "quad":
            # the same launch without its position stores (okx_solve_opts.output = none): what the records cost
            n_launch = dp.plan(targets, info_out=info, chain_len=chain_len, predictor=False, output="none", **kw)
            n_wall, n_ms = time_launches(n_launch, steps, warmup, device)
            res[tag]["output_none"] = {"value": n / n_wall, "kernel_ms": n_ms}
        if res[tag]["kernel"].startswith("lane") and not os.environ.get("OKX_BENCH_NO_QUAD_COMPARE"):
            # the quad kernel on the same launch, for the record (what round 2 measured)
            q_launch = dp.plan(targets, out=out, info_out=info, chain_len=chain_len, predictor=False, kernel="quad", **kw)
            q_wall, q_ms = time_launches(q_launch, steps, warmup, device)
            res[tag]["quad_kernel"] = {"value": n / q_wall, "kernel_ms": q_ms, "lm_evaluations_mean": info_summary(info)[0]}
    if roles_geometry is not None and dp.kernel == "quad":
        try:
            measure = measure_evaluated_axle if "axle" in os.path.basename(roles_geometry) else measure_evaluated
            res["evaluated"] = measure(dp, roles_geometry, targets, out, info, kw, device, steps, warmup)
        except Exception as exc:  # (an extra leg must not take the line down with it)
            res["evaluated"] = {"error": f"{type(exc).__name__}: {exc}"}
    if "cold" in res:
        res["cold"]["start"] = "every problem an independent cold start from its geometry's design state (SURVEY.md section 8d)"
    if "chained" in res:
        res["chained"]["start"] = ("chain_len=-1: one warm-started chain per resident problem slot, secant / quadratic "
                                   "extrapolation along the chain (reference warm start, solver.py:774); no fitted model")
    dp.close()
    return res


def measure_evaluated_axle(dp, suspension_yaml: str, targets, out, info, kw: dict, device, steps: int, warmup: int) -> dict:
    """What an EVALUATED state of a COMPOSED AXLE costs (reference core/sweep.py:217-270 for an AxleSuspension): the solve, its
    solution-manifold tangents, both corners' metric catalogs with their derivative columns, the axle-scope metrics and the
    rotation roles - as six launches (solve, tangents, 2 x corner metrics, axle metrics, rotations) and as ONE
    (okx_solve_evaluated_batch on the pair-mode evaluated module)."""
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import (axis_rotation_metrics, axle_evaluation_roles, axle_roles, axle_state_metrics,
                                             corner_state_metrics, topology_rotation_roles)

    program = dp.program
    axle = load_geometry(suspension_yaml)
    roles, rot_names, hw_names = axle_evaluation_roles(axle, program)
    dp.enable_evaluation(roles)
    columns = dp.eval_columns
    n, T = targets.shape[0], program.n_targets
    evb = torch.empty((n, 1 + T, columns), dtype=torch.float64, device=device)
    skw = dict(chain_len=1, predictor=False, **kw)
    solve = dp.plan(targets, out=out, info_out=info, **skw)
    _, solve_ms = time_launches(solve, steps, warmup, device)
    solve()
    tan, _ = dp.tangents(out)
    left, right = axle_roles(axle, program)
    names, rroles = topology_rotation_roles(axle, program)
    _, tan_ms = time_launches(lambda: dp.tangents(out), steps, warmup, device)
    _, met_ms = time_launches(lambda: (corner_state_metrics(left, out, tan), corner_state_metrics(right, out, tan)), steps, warmup, device)
    _, axle_ms = time_launches(lambda: axle_state_metrics(left, right, out), steps, warmup, device)
    rot_ms = time_launches(lambda: axis_rotation_metrics(rroles, out, tan), steps, warmup, device)[1] if names else 0.0
    del tan
    fused = dp.plan_evaluated(targets, info_out=info, eval_out=evb, output="none", **skw)
    wall, fused_ms = time_launches(fused, steps, warmup, device)
    nfev, ok = info_summary(info)
    _, fused_rec_ms = time_launches(dp.plan_evaluated(targets, out=out, info_out=info, eval_out=evb, **skw), steps, warmup, device)
    given_ms = time_launches(lambda: dp.evaluate(out, eval_out=evb), steps, warmup, device)[1]
    flags = evb[:, 0, 21]
    bytes_out = 8 * T + 8 * columns * (1 + T) + 16
    separate = solve_ms + tan_ms + met_ms + axle_ms + rot_ms
    return {"unit": "evaluated states/s", "value": n / (fused_ms * 1e-3), "kernel_ms": fused_ms, "wall_value": n / wall,
            "kernel": "pair mode: one quad per half; lane c of each quad evaluates direction c - 1 of its corner's catalog, the left "
                      "quad the axle-scope metrics, both the rotation roles",
            "roles": sorted(set(rot_names) | set(hw_names)),
            "with_records_kernel_ms": fused_rec_ms, "given_states_kernel_ms": given_ms,
            "separate_launches": {"solve_ms": solve_ms, "tangents_ms": tan_ms, "two_corner_metrics_with_derivatives_ms": met_ms,
                                  "axle_metrics_ms": axle_ms, "rotation_roles_ms": rot_ms, "total_ms": separate,
                                  "value": n / (separate * 1e-3)},
            "speedup": separate / fused_ms, "all_converged": ok, "lm_evaluations_mean": nfev,
            "tangent_solves_ok": bool((flags == 1.0).all().item()),
            "algorithmic_bytes_per_state": bytes_out, "algorithmic_gbs": bytes_out * n / (fused_ms * 1e-3) / 1e9,
            "hbm_frac": bytes_out * n / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": f"HIP events; output = none: 8 T in, {columns} (1 + T) doubles + 16 B of info out per state; issue-bound like the solve"}


def measure_evaluated(dp, suspension_yaml: str, targets, out, info, kw: dict, device, steps: int, warmup: int) -> dict:
    """What an EVALUATED state costs (reference core/sweep.py:217-270, solve_evaluated_sweep for a batch): the solve, its
    solution-manifold tangents and the corner metric catalog with every derivative column - as three launches (the state
    records written by the solve and re-read twice, the tangents written and re-read) and as ONE (okx_solve_evaluated_batch:
    tangents and metrics are the solve kernel's epilogue, taken at the converged state while it is in registers;
    output = none: nothing but the info records and the [1 + T][24] evaluation rows leaves the chip)."""
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles, corner_state_metrics

    program = dp.program
    roles = corner_roles(load_geometry(suspension_yaml), program)
    dp.enable_evaluation(roles)
    n, T = targets.shape[0], program.n_targets
    evb = torch.empty((n, 1 + T, 24), dtype=torch.float64, device=device)
    skw = dict(chain_len=1, predictor=False, **kw)
    tkw = {k: v for k, v in kw.items() if k in ("geom_pos", "geom_row_param", "steps_per_geometry")}
    solve = dp.plan(targets, out=out, info_out=info, **skw)
    _, solve_ms = time_launches(solve, steps, warmup, device)
    solve()
    tan, _ = dp.tangents(out, **tkw)
    _, tan_ms = time_launches(lambda: dp.tangents(out, **tkw), steps, warmup, device)
    _, met_ms = time_launches(lambda: corner_state_metrics(roles, out, tan), steps, warmup, device)
    del tan
    fused = dp.plan_evaluated(targets, info_out=info, eval_out=evb, output="none", **skw)
    wall, fused_ms = time_launches(fused, steps, warmup, device)
    nfev, ok = info_summary(info)
    _, fused_rec_ms = time_launches(dp.plan_evaluated(targets, out=out, info_out=info, eval_out=evb, **skw), steps, warmup, device)
    flags = evb[:, 0, 21]
    bytes_out = 8 * T + 8 * 24 * (1 + T) + 16
    return {"unit": "evaluated states/s", "value": n / (fused_ms * 1e-3), "kernel_ms": fused_ms, "wall_value": n / wall,
            "kernel": "lane form (one lane per problem, duals with all target directions)" if dp.evaluation & 2 and n >= max(dp.lane_threshold, 1)
                      else "quad form (lane c of a quad evaluates direction c - 1)",
            "with_records_kernel_ms": fused_rec_ms,
            "three_launches": {"solve_ms": solve_ms, "tangents_ms": tan_ms, "metrics_with_derivatives_ms": met_ms,
                               "total_ms": solve_ms + tan_ms + met_ms, "value": n / ((solve_ms + tan_ms + met_ms) * 1e-3)},
            "speedup": (solve_ms + tan_ms + met_ms) / fused_ms, "all_converged": ok, "lm_evaluations_mean": nfev,
            "tangent_solves_ok": bool((flags == 1.0).all().item()),
            "algorithmic_bytes_per_state": bytes_out, "algorithmic_gbs": bytes_out * n / (fused_ms * 1e-3) / 1e9,
            "hbm_frac": bytes_out * n / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "HIP events; output = none: 8 T in, 24 (1 + T) doubles + 16 B of info out per state (the three-launch path moves "
                    "24 n_out (2 + 2 T) + 152 (1 + T) B per state through HBM); issue-bound like the solve"}


def measure_downstream(dp, suspension_yaml: str, positions, device, reps: int = 20, sweep_targets=None) -> dict:
    """The callers after the solve on the same 16384 solved states, each one launch: solution-manifold
    tangents (okx_tangent_batch) and the corner metric catalog with its derivative columns (okx_corner_metrics_batch)."""
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles, corner_state_metrics

    program = dp.program
    roles = corner_roles(load_geometry(suspension_yaml), program)
    n = positions.shape[0]

    def timed(fn):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / reps, out

    t_tan, (tangents, _) = timed(lambda: dp.tangents(positions))
    t_met, _ = timed(lambda: corner_state_metrics(roles, positions, tangents))

    # ... and at C5 scale (a million states: 377 MB of records, past the Infinity Cache), HIP-event time of each kernel and its
    # algorithmic GB/s: the kernels beside the solve are the streaming ones (tools/stream_rates.py, profiles/r04/EXPERIMENTS.md 9)
    def streaming(n_big: int = 1 << 20) -> dict:
        reps_of = -(-n_big // n)
        big = positions.repeat(reps_of, 1, 1)[:n_big].contiguous()
        free = big[:, dp.free_out_index].contiguous()
        out = torch.empty_like(big)

        def ev(fn, k=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(k):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) / k

        T, n_out, n_free = program.n_targets, program.n_out, program.n_free
        rows = {}

        def row(name, ms, bytes_per_state):
            rows[name] = {"ms": ms, "states_per_s": n_big / ms * 1e3, "bytes_per_state": bytes_per_state,
                          "algorithmic_gbs": bytes_per_state * n_big / ms / 1e6, "hbm_frac": bytes_per_state * n_big / ms / 1e6 / HBM_PEAK_GBS}

        row("expand_free_to_records", ev(lambda: dp.expand(free, out=out)), 24 * n_free + 24 * n_out)
        row("copy_of_the_records_reference", ev(lambda: out.copy_(big)), 48 * n_out)
        tan_big, _ = dp.tangents(big)
        row("tangents", ev(lambda: dp.tangents(big)), 24 * n_out * (1 + T) + 24)
        row("corner_metrics", ev(lambda: corner_state_metrics(roles, big, None)), 24 * n_out + 152)
        row("corner_metrics_with_derivatives", ev(lambda: corner_state_metrics(roles, big, tan_big)), 24 * n_out * (1 + T) + 152 * (1 + T))
        del tan_big
        # ... and the same evaluation as the solve kernel's epilogue: one launch from targets to metric / derivative rows
        # (the sweep's targets tiled to a million problems of the own geometry; the ensemble form - per-geometry tables,
        # BASELINE config 5 itself - is other_configs[C5].evaluated)
        try:
            dp.enable_evaluation(roles)
            big_t = sweep_targets.repeat(reps_of, 1)[:n_big].contiguous()
            info_big = torch.empty((n_big, 40), dtype=torch.uint8, device=device)
            evb = torch.empty((n_big, 1 + T, 24), dtype=torch.float64, device=device)
            solve_ms = ev(dp.plan(big_t, out=out, info_out=info_big, chain_len=1, predictor=False))
            row("solve_records", solve_ms, 8 * T + 24 * n_out + 16)
            row("evaluated_one_launch", ev(dp.plan_evaluated(big_t, info_out=info_big, eval_out=evb, output="none", chain_len=1, predictor=False)),
                8 * T + 8 * 24 * (1 + T) + 16)
            three = solve_ms + rows["tangents"]["ms"] + rows["corner_metrics_with_derivatives"]["ms"]
            rows["evaluated_three_launches"] = {"ms": three, "states_per_s": n_big / three * 1e3}
            # okx_evaluate_batch: tangents + metrics + derivative columns of GIVEN states in one launch (its lane form here)
            row("evaluate_given_states", ev(lambda: dp.evaluate(out, eval_out=evb)), 24 * n_out + 8 * 24 * (1 + T))
            rows["evaluated_one_launch"]["speedup_over_three_launches"] = three / rows["evaluated_one_launch"]["ms"]
        except Exception as exc:  # noqa: BLE001
            rows["evaluated_one_launch"] = {"error": f"{type(exc).__name__}: {exc}"}
        return {"states": n_big, "rows": rows,
                "note": "HIP events around 10 launches each, the sweep's states tiled to a million; bound: HBM for expand / corner_metrics "
                        "(a torch copy of the records is the practical ceiling), fp64 issue for tangents and the derivative columns"}

    try:
        c5_scale = streaming()
    except Exception as exc:  # (an extra leg must not take the headline down with it)
        c5_scale = {"error": f"{type(exc).__name__}: {exc}"}
    return {"states": n, "c5_scale": c5_scale,
            "tangents": {"value": n / t_tan, "unit": "states/s", "ms": t_tan * 1e3,
                         "bytes_per_state": 24 * program.n_out * (1 + program.n_targets) + 24},
            "corner_metrics_with_derivatives": {"value": n / t_met, "unit": "states/s", "ms": t_met * 1e3,
                                                "bytes_per_state": 24 * program.n_out * (1 + program.n_targets) + 152 * (1 + program.n_targets)},
            "note": "wall time per call incl. the host side of the C-ABI call and output allocation; inputs and outputs in HBM"}


def measure_dropin(device) -> dict:
    """The reference's own entry point on its own benchmark workloads (tests/benchmarks/test_bench_sweep.py:29-40
    times `solve_sweep` on the rocker axle; BASELINE config 1 is the 101-step bump sweep): wall-clock per call."""
    from open_kinematics_amd.input import build_sweep, load_geometry, load_sweep
    from open_kinematics_amd.solver import clear_program_cache
    from open_kinematics_amd.sweep import solve_sweep
    from open_kinematics_amd.workloads import geometry_path

    import yaml

    out = {}
    clear_program_cache()
    cases = []
    sus = load_geometry(geometry_path("geometry.yaml"))
    with open(geometry_path("bump_sweep.yaml"), "r", encoding="utf-8") as fh:
        mapping = yaml.safe_load(fh)
    mapping["steps"] = 101  # BASELINE config 1: the file ships 36 steps
    cases.append(("c1_101_steps", sus, build_sweep(mapping, sus)))
    axle = load_geometry(geometry_path("axle_geometry_rocker.yaml"))
    cases.append(("rocker_axle_sweep", axle, load_sweep(geometry_path("axle_rocker_sweep.yaml"), axle)))
    for name, suspension, sweep in cases:
        times = []
        for _ in range(6):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            states, infos = solve_sweep(suspension, sweep, device=device)
            times.append((time.perf_counter() - t0) * 1e3)
        out[f"{name}_first_call_ms"] = times[0]
        out[f"{name}_ms"] = float(np.median(times[1:]))
        out[f"{name}_steps"] = len(states)
        out[f"{name}_nfev_mean"] = float(np.mean([i.nfev for i in infos]))
    out["note"] = ("solve_sweep(suspension, sweep) -> (states, infos), host objects in and out; first call = flatten + "
                   "program upload + kernel generation + code-object load (in-tree cache) + solve, later calls reuse "
                   "the cached device program (solver._PROGRAM_CACHE); a sweep of four steps and more is solved as cold "
                   "starts side by side and kept when it is the sequential warm start's path (one chain otherwise): a "
                   "latency figure, not a throughput figure")
    clear_program_cache()
    return out


# --------------------------------------------------------------------------------------------------
# main
# --------------------------------------------------------------------------------------------------

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~11 ms of warm-up launches (the clocks ramp over milliseconds) and ~45 ms of timed ones per rank
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--config", choices=("c2", "c5"), default="c2",
                    help="c2: BASELINE config 2, weak scaling (default, the headline); c5: BASELINE config 5, 4096 perturbed "
                         "geometries x 256 steps, geometry-major shards (strong scaling)")
    ap.add_argument("--preheat-ms", type=float, default=40.0,
                    help="milliseconds of untimed launches ahead of the warm-up steps (GPU clocks leave their idle state); 0 = none")
    ap.add_argument("--torch-submit", action="store_true",
                    help="one GPU, graph submission: record / launch / record / poll through torch's own calls instead of the HIP runtime directly")
    ap.add_argument("--no-graph", action="store_true",
                    help="one GPU: submit the K timed steps as K stream launches instead of ONE HIP graph of K kernel nodes "
                         "(measured on a quiet host: 16.09 against 16.30 us per step at K = 2000, no difference at K = 20; the graph "
                         "keeps 0.3 ms of GPU work independent of the host's launch loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5-sharded", action="store_true",
                    help="leave out the c5_sharded leg (BASELINE config 5 through dist.ShardedEnsemble in this process group)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip with_model / e2e / other_configs / dropin (what the profiling runs use)")
    ap.add_argument("--start", choices=("design", "model"), default="design",
                    help="design: cold starts from the design state (section 8d, the headline); model: the fitted chain-head model "
                         "(profiling runs of the predictor path; never the default)")
    ap.add_argument("--chain-len", type=int, default=-1, help="-1 auto (16384 steps fit the chip: independent solves), 1 = independent")
    ap.add_argument("--c5-chunks", type=int, default=0, help="--config c5, N > 1: chunks of the pipelined exchange per rank (0: auto, up to 8)")
    ap.add_argument("--c5-gather", choices=("records", "free", "metrics"), default="records",
                    help="--config c5, N > 1: every rank ends with all output records (default), with the free coordinates only (no expand), "
                         "or - the EVALUATED ensemble - with four metric columns of every state (camber, camber gain and bump steer "
                         "along the bump target, the hub's rise rate): each rank evaluates its shard, 33 bytes per state travel")
    ap.add_argument("--c5-info", choices=("full", "status"), default="status",
                    help="--config c5, N > 1: what travels beside the coordinates - the 40-byte info records or one status byte per solve")
    ap.add_argument("--rccl-world-one", action="store_true",
                    help="one rank, but with an RCCL process group and the all-gather in the step: what a one-GPU box can "
                         "rehearse of the N > 1 path (communicator, stream ordering of the pipeline, expand of the gathered block)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher plumbing only: start the ranks, have rank 0 print a line, touch no GPU (CPU test)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank uses cuda:0 and the exchange runs over gloo "
                         "(exercises the sharding / pipeline / rebuild logic, not RCCL; the numbers mean nothing)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as children, BEFORE anything here
        # has touched the GPU (a process that has initialised HIP must never exec / re-exec), hand their output
        # through (rank 0 prints the JSON line) and leave with the launcher's return code.
        sys.exit(self_launch(args.gpus, sys.argv[1:], dry=args.dry_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.dry_launch:
        # launcher plumbing only (a CPU test drives this): no GPU, no process group
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": world,
                              "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the solve path)")
    host_threads = fit_host_threads(processes=world)  # before the first parallel host op (hostcpu.py: the cgroup quota)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.rccl_world_one:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                free_port = sock.getsockname()[1]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    if args.config == "c5":
        line = run_c5(args, world, rank, device)
    else:
        line = run_c2(args, world, rank, device)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1 or args.rccl_world_one:
        if getattr(run_c2, "leg_timed_out", False):
            os._exit(0)  # (a rank is stuck in an exchange: no further collective, the line is out)
        # the closing barrier under a deadline too: a peer that gave up on the c5 leg has left without it
        _, late = _with_deadline(lambda: (dist.barrier(), dist.destroy_process_group()) and None, 60.0 if world > 1 else None, device)
        if late:
            os._exit(0)


def self_launch(n_ranks: int, argv: list, dry: bool = False) -> int:
    """Run this script under `python -m torch.distributed.run` with one rank per GPU of this node (rendezvous on
    127.0.0.1, a free port) and return the launcher's exit code.  The ranks inherit stdout / stderr."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dry:
        env["HIP_VISIBLE_DEVICES"] = ""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _direct_hip_submission(graph, stream, start_event, end_event):
    """
    Handles for submitting a captured graph through the HIP runtime this process already maps (torch's own copy: a second
    copy of the library would not know torch's events and graphs): (``(hip, graph_exec, stream, ev0, ev1)``, None), or
    (None, reason) when this torch / ROCm does not expose them - the caller then uses torch's calls and says so in the line.
    """
    import ctypes as C

    try:
        with open("/proc/self/maps", "r", encoding="utf-8") as maps:
            mapped = sorted({line.split()[-1] for line in maps if "libamdhip64" in line})
        if len(mapped) != 1:
            return None, f"{len(mapped)} HIP runtimes mapped"
        hip = C.CDLL(mapped[0])
        hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        hip.hipEventQuery.argtypes = [C.c_void_p]
        return (hip, C.c_void_p(graph.raw_cuda_graph_exec()), C.c_void_p(stream.cuda_stream),
                C.c_void_p(start_event.cuda_event), C.c_void_p(end_event.cuda_event)), None
    except Exception as exc:  # noqa: BLE001 - a torch without the raw handles
        return None, f"{type(exc).__name__}: {exc}"


def timed_region(step, drain, steps: int, warmup: int, world: int, device, per_launch_events: bool, preheat_ms: float = 0.0,
                 graph_steps: bool = False):
    """The contract's timed region: W warm-up steps, barrier + synchronize, K steps, synchronize + barrier, MAX over
    ranks.  `step(k, start_event, end_event)` records the events around its solve launch when they are given.
    `preheat_ms` > 0: the same step is launched for that long ahead of the W warm-up steps (reported in the line as
    `preheat`): a GPU that has been idle runs its first few milliseconds of kernels at reduced clocks - 20 steps after 5
    warm-up steps measured 8 % slower than the same kernel after 11 ms of launches (profiles/r03/EXPERIMENTS.md section 10).
    `graph_steps` (one rank, no exchange; the current stream must not be the default stream): the K timed steps are captured
    ahead of the timed region into ONE HIP graph of K kernel nodes and the timed region launches that graph - the same K
    kernels, submitted in one call, so that 0.3 ms of GPU work are not at the mercy of the host's launch loop."""
    # The graph is captured FIRST: capture and instantiation are host work during which the GPU idles (milliseconds) - after
    # the preheat they would hand the timed region a GPU that is leaving idle again (kernel_ms 17.2 instead of 16.1 us).
    graph = None
    timed_region.graph_note = None
    if graph_steps and world == 1 and not per_launch_events:
        step(0, None, None)  # (whatever a first launch sets up - first-step tables, plans - happens outside the capture)
        drain()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=torch.cuda.current_stream(device)):
                for k in range(steps):
                    step(k, None, None)
            graph.replay()  # (the first launch of a graph uploads it: untimed)
            drain()
        except Exception as exc:  # a runtime that refuses the capture: the K steps are launched one by one, and the line says so
            graph = None
            timed_region.graph_note = f"graph capture failed ({type(exc).__name__}: {exc}); stream launches"
            torch.cuda.synchronize(device)
    timed_region.graph = graph is not None
    n_pairs = steps if per_launch_events else 1
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_pairs)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_pairs)]
    # torch creates the HIP event at its first record(), and the process's first timing event initialises the runtime's
    # timestamp machinery (~45 us, measured): both happen here, ahead of the preheat, not inside the K timed steps and not
    # between the warm-up and them
    for ev in starts + ends:
        ev.record()
    preheat_steps = 0
    if preheat_ms > 0.0:
        torch.cuda.synchronize(device)
        if world > 1:
            # several ranks: a FIXED count (every step is a collective call - the ranks must make the same number of them)
            for _ in range(int(preheat_ms * 50)):  # ~20 us per step
                step(preheat_steps, None, None)
                preheat_steps += 1
        else:
            t_end = time.perf_counter() + preheat_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(32):
                    step(preheat_steps, None, None)
                    preheat_steps += 1
    timed_region.preheat_steps = preheat_steps
    for k in range(warmup):
        step(k, None, None)
    if graph is not None:
        graph.replay()  # (untimed: the launch that precedes the timed one is of the same kind - a graph launch after
        #                  thousands of stream launches took 50 - 80 us to reach the GPU, 3 us per step at K = 20)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    # Graph submission through the HIP runtime directly (the same three calls torch makes - record, launch, record - and
    # the poll, without torch's per-call Python layers: ~15 us less host time inside a 0.35 ms region; --torch-submit: off)
    direct, timed_region.direct_note = None, None
    if graph is not None and not per_launch_events and not getattr(timed_region, "torch_submit", False):
        direct, timed_region.direct_note = _direct_hip_submission(graph, torch.cuda.current_stream(device), starts[0], ends[0])
    timed_region.direct = direct is not None
    t0 = time.perf_counter()
    if direct is not None:
        hip, exec_h, stream_h, ev0, ev1 = direct
        rc = hip.hipEventRecord(ev0, stream_h) or hip.hipGraphLaunch(exec_h, stream_h) or hip.hipEventRecord(ev1, stream_h)
        if rc != 0:
            raise RuntimeError(f"HIP error {rc} submitting the timed graph")
        t_sub = time.perf_counter()
        rc = hip.hipEventQuery(ev1)
        while rc == 600:  # hipErrorNotReady; anything else that is not success is a device error: raise, never spin on it
            rc = hip.hipEventQuery(ev1)
        if rc != 0:
            raise RuntimeError(f"HIP error {rc} while waiting for the timed graph")
        t_poll = time.perf_counter()
        timed_region.breakdown = [t_sub - t0, t_poll - t_sub]
    else:
        if not per_launch_events:
            starts[0].record()
        if graph is not None:
            graph.replay()
        else:
            for k in range(steps):
                step(k, starts[k] if per_launch_events else None, ends[k] if per_launch_events else None)
        if not per_launch_events:
            ends[0].record()
            # (the host polls the end event before it synchronises: a poll sees the end of the GPU's work within microseconds, an
            #  interrupt-driven wait wakes the host later; the synchronize below then finds an idle GPU - profiles/r04/EXPERIMENTS.md section 10)
            while not ends[0].query():
                pass
    drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if getattr(timed_region, "breakdown", None) and len(timed_region.breakdown) == 2:
        timed_region.breakdown.append(elapsed - sum(timed_region.breakdown))   # drain + synchronize
    kernel_ms = float(np.sum([s.elapsed_time(e) for s, e in zip(starts, ends)])) / steps
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0].item()), float(t[1].item())
    return elapsed, kernel_ms


def run_c2(args, world: int, rank: int, device) -> dict:
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.dist import FreeGatherPipeline, shard_range
    from open_kinematics_amd.workloads import (axle_grid_problem, bump_sweep_problem, ensemble_problem,
                                               macpherson_grid_problem)

    n_total = STEPS_PER_RANK * world
    program, targets_all = bump_sweep_problem(n_total)
    lo, hi = shard_range(n_total, rank, world)
    use_graph = world == 1 and not args.rccl_world_one and not args.no_graph
    timed_region.torch_submit = bool(args.torch_submit)
    if world == 1:
        # one GPU: everything below (plans, events, the extra legs) runs on a stream of its own - stream capture, which the
        # graph submission of the timed steps needs, is not allowed on the default stream
        torch.cuda.set_stream(torch.cuda.Stream(device))
    dp = DeviceProgram(program, device)
    targets = torch.as_tensor(targets_all[lo:hi], device=device).contiguous()
    info = torch.empty((hi - lo, 40), dtype=torch.uint8, device=device)
    # Two output slots: with N > 1 ranks the all-gather of step k (RCCL stream) overlaps the solve of step k + 1
    # (launch stream), dist.GatherPipeline.  The exchange ships the free coordinates of each solve (144 B) and every
    # rank rebuilds the full positions (360 B) itself (dist.FreeGatherPipeline / okx_expand_positions_batch).
    pipe = FreeGatherPipeline(hi - lo, program.n_out, program.n_free, dp.expand, torch.float64, device,
                              collective_at_world_one=args.rccl_world_one)
    use_model = args.start == "model" and bool(dp.fit_predictor(targets))
    cold_kw = dict(chain_len=args.chain_len, predictor=use_model)
    # pre-bound launches: per step the host only makes the C-ABI call
    # (with an exchange the solve writes the free coordinates - the payload - straight into the send buffer)
    launches = [dp.plan(targets, out=buf, info_out=info, output=pipe.output, **cold_kw) for buf in pipe.solve_buffers]

    def step(k, start, end):
        pipe.begin(k)
        if start is not None:
            start.record()
        launches[k % len(launches)]()
        if end is not None:
            end.record()
        pipe.submit(k)

    # One rank: a single event pair brackets the K back-to-back launches (average = elapsed / K, launch gaps included;
    # per-launch pairs would add two stream markers per ~30 us kernel).  Several ranks: per-launch pairs, so the
    # exchange between launches is excluded from the kernel time.
    exchanging = world > 1 or args.rccl_world_one
    elapsed, kernel_ms = timed_region(step, pipe.drain, args.steps, args.warmup, world, device, per_launch_events=exchanging,
                                      preheat_ms=args.preheat_ms, graph_steps=use_graph)
    preheat_steps = timed_region.preheat_steps
    if args.rccl_world_one:
        # the gathered + expanded block of the last step must be the locally solved one, bit for bit
        gathered = pipe.drain()
        records = dp.solve(targets, **cold_kw).positions
        if not torch.equal(gathered, records):
            raise SystemExit("rccl-world-one: the gathered and re-expanded positions differ from the solver's own records")
    nfev_mean, ok = info_summary(info)
    # BASELINE config 5 through the sharded pipeline, in this process group (every rank takes part): the workload of north_star
    # that is "sharded across 8 x MI355X" rides in the default line at every N, beside the xGMI-bound per-step gather of C2
    c5_sharded = None
    if not args.no_c5_sharded and not args.rccl_world_one:
        # (under a deadline when ranks exchange: the headline above is measured; a point-to-point exchange that never
        #  completes on some node must cost this leg, not the line - main() then leaves without another collective)
        c5_sharded, timed_out = _with_deadline(lambda: measure_c5_sharded(world, rank, device), C5_LEG_DEADLINE_S if world > 1 else None, device)
        run_c2.leg_timed_out = timed_out
    if rank != 0:
        return {}

    stats = plan_stats(program)
    bytes_per_solve = algorithmic_bytes_per_solve(program)
    default_run = world == 1 and args.chain_len == -1 and dp.kernel == "quad" and not args.rccl_world_one
    traffic, traffic_src = committed_traffic("bench_c2_cold" if not use_model else "bench_c2_model") if default_run else (None, None)
    achieved_gbs = bytes_per_solve * (hi - lo) / (kernel_ms * 1e-3) / 1e9
    flops = estimated_flops_per_evaluation(program, stats) * nfev_mean * (hi - lo)
    free_bytes = STEPS_PER_RANK * program.n_free * 24
    line = {
        "metric": "constraint solves/sec (sweep steps/sec), double-wishbone bump sweep",
        "value": n_total * args.steps / elapsed,
        "unit": "constraint solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preheat": {"ms": args.preheat_ms, "steps": preheat_steps,
                    "note": "untimed launches of the same step ahead of the W warm-up steps, so that the K timed steps run at the "
                            "clocks of a GPU under load rather than of one leaving idle (--preheat-ms 0 switches it off)"},
        "submission": ({"mode": "hip graph", "note": "the K timed steps are K kernel nodes of ONE HIP graph, captured ahead of the timed "
                        "region and launched once inside it (--no-graph: K stream launches, as `sustained` below)",
                        "host_us": dict(zip(("submit", "poll_until_done", "drain_and_synchronize"),
                                            [round(x * 1e6, 1) for x in getattr(timed_region, "breakdown", None) or []])),
                        "calls": "hipEventRecord / hipGraphLaunch / hipEventRecord / hipEventQuery through the HIP runtime directly"
                                 if getattr(timed_region, "direct", False) else "torch.cuda.Event.record / CUDAGraph.replay / Event.query",
                        **({"direct_fallback": timed_region.direct_note} if getattr(timed_region, "direct_note", None) else {})}
                       if getattr(timed_region, "graph", False) else
                       {"mode": "stream launches", **({"note": timed_region.graph_note} if getattr(timed_region, "graph_note", None) else {})}),
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "double-wishbone corner (tests/data/geometry.yaml), 16384-step fp64 bump sweep "
                        "-60..+80 mm per GPU, rack held (BASELINE config 2)",
            "problems_per_gpu": STEPS_PER_RANK,
            "n_vars": program.n_vars,
            "n_residual_rows": program.n_residuals,
            "line_mode": program.line_mode,
            "kernel": dp.kernel,
            "start": ("fitted chain-head model (--start model; NOT the section 8d rule)" if use_model else
                      "every one of the 16384 solves per GPU is an independent cold start from the design state "
                      "(SURVEY.md section 8d; reference solver.py:710): chain_len=-1 resolves to chains of length 1 because the "
                      "16384 steps fit the chip's 16384 resident problem slots.  The Levenberg-Marquardt pass AT the design "
                      "state (Jacobian, J^T J, damped factorisation) is identical for every problem of a geometry, so it is "
                      "evaluated once per geometry (okx_quad_head_u: one wavefront at program set-up, cached per lambda0) "
                      "and each problem takes its first step from that table, then iterates on its own "
                      "(okx_solve_opts.shared_first_step, default on; the same launch with every problem running its own "
                      "first pass is reported under own_first_pass)"),
            "lm_evaluations_mean": nfev_mean,
            "lm_evaluations_note": "evaluations each problem ran itself; the shared design-state pass is not counted",
            "predictor": bool(use_model),
            "all_converged": ok,
            "exchange": ("RCCL all-gather of the solved free coordinates every step (every rank rebuilds all positions from "
                         "them), overlapped with the next step's solve (two output slots); %d B per rank per step instead of "
                         "%d B of positions; every rank receives %d B per step (DESIGN.md section 8)"
                         % (free_bytes, STEPS_PER_RANK * program.n_out * 24, (world - 1) * free_bytes))
                        if exchanging else "none",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_measured_in_this_run": False if traffic is not None else None,
            "kernel": ("okx_lane_solve_u" if 0 < dp.lane_threshold <= (hi - lo) and (dp.lane_bodies & 1) else
                       ("okx_quad_cold_u" if dp.has_cold_body and not use_model and args.chain_len in (-1, 1) else "okx_quad_solve_u")
                       if dp.kernel == "quad" else "okx_solve_kernel"),
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_solve": bytes_per_solve,
            "note": "HBM is the bound north_star states and frac is reported against it, but at 392 B per solve the path is "
                    "fp64-issue / latency bound (SURVEY.md section 8d): see compute",
        },
        "compute": {
            "unit": "TFLOP/s",
            "achieved": flops / (kernel_ms * 1e-3) / 1e12,
            "peak": FP64_VECTOR_PEAK_TFLOPS,
            "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "flops_per_solve_estimate": flops / (hi - lo),
            "source": "analytic flop model x measured LM evaluations (not a counter reading)",
        },
    }
    counted, counted_src = committed_fp64("bench_c2_cold" if not use_model else "bench_c2_model") if default_run else (None, None)
    if counted:
        hw = counted["flops_64_lanes"] / (kernel_ms * 1e-3) / 1e12
        line["compute"]["counters"] = {
            "fp64_flops_per_launch_64_lanes": counted["flops_64_lanes"], "achieved": hw, "frac": hw / FP64_VECTOR_PEAK_TFLOPS,
            "useful_frac": 0.75 * hw / FP64_VECTOR_PEAK_TFLOPS, "source": counted_src, "measured_in_this_run": False,
            "note": "SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 of the committed PMC pass x 64 lanes over this run's kernel time; one lane in "
                    "four carries zeros in the quad layout (useful_frac = 3/4)"}
    if exchanging:
        line["solve_only"] = {
            "value": n_total / (kernel_ms * 1e-3),
            "kernel_ms_max_over_ranks": kernel_ms,
            "note": "all ranks' solves / the slowest rank's mean solve-kernel time (per-launch HIP events): how the solve "
                    "itself scales, exchange excluded; `value` above is exchange-inclusive",
        }
        line["exchange"] = {"bytes_sent_per_rank_per_step": free_bytes, "bytes_received_per_rank_per_step": (world - 1) * free_bytes,
                            "collective": "all_gather_into_tensor (RCCL)" if not args.rehearse_on_one_gpu else "gloo (rehearsal)"}
    if world == 1 and not args.rccl_world_one:
        # Sustained figure: a few thousand back-to-back launches (the 20-step driver run times 0.6 ms of GPU work).
        sus_wall, sus_ms = time_launches(launches[0], 2000, 10, device)
        line["sustained"] = {"value": (hi - lo) / sus_wall, "kernel_ms": sus_ms, "launches": 2000,
                             "note": "the same launch 2000 times back to back on one stream, one HIP-event pair around them"}
    if world == 1 and not args.no_extras and not args.rccl_world_one:
        line["one_shot"] = measure_one_shot(program, targets, device, kernel_ms)
        extra_steps = max(5, min(args.steps, 50))
        own = dp.plan(targets, out=pipe.solve_buffers[0], info_out=info, chain_len=args.chain_len, predictor=False, shared_first_step=False)
        own_wall, own_ms = time_launches(own, args.steps, args.warmup, device)
        own_nfev, own_ok = info_summary(info)
        line["own_first_pass"] = {"value": (hi - lo) / own_wall, "kernel_ms": own_ms, "lm_evaluations_mean": own_nfev,
                                  "all_converged": own_ok,
                                  "note": "shared_first_step=0: every problem evaluates the design state itself (round-1 behaviour)"}
        line["pipelined"] = measure_pipelined(dp, targets, device, max(args.steps, 60))
        line["with_model"] = measure_with_model(program, targets, device, args.steps, args.warmup)
        line["e2e"] = measure_e2e(dp, targets_all[lo:hi], device, max(extra_steps, 200), dict(chain_len=args.chain_len, predictor=False))
        line["e2e"]["host_threads"] = fit_host_threads()  # (what main() decided: the first call's answer is kept)
        line["e2e"]["compact"] = measure_e2e_compact(dp, targets_all[lo:hi], device, max(extra_steps, 200),
                                                     dict(chain_len=args.chain_len, predictor=False))
        # (last of the host-to-host legs on purpose: round 3 measured this leg at a fifth of its rate whenever another leg had
        #  run before it; the cause - 24-byte store fragments whose merging on the way to the host depended on timing - is
        #  gone with the cold body's staged stores, profiles/r04/EXPERIMENTS.md section 5)
        line["e2e"]["zero_copy"] = measure_e2e_zero_copy(dp, targets_all[lo:hi], device, 200, dict(chain_len=args.chain_len, predictor=False))
        from open_kinematics_amd.workloads import geometry_path
        line["downstream"] = measure_downstream(dp, geometry_path("geometry.yaml"), pipe.solve_buffers[0], device, sweep_targets=targets)
        dp.close()
        line["other_configs"] = [
            measure_config("C3 rocker + U-bar axle, 256x256 heave x roll grid (n = 60, pair mode)",
                           lambda: axle_grid_problem(256, 256), device, 20, 3, roles_geometry=geometry_path("axle_geometry_rocker.yaml")),
            measure_config("C4 MacPherson corner, 512x512 bump x rack grid", lambda: macpherson_grid_problem(512, 512), device, 20, 3,
                           roles_geometry=geometry_path("macpherson_geometry.yaml")),
            measure_config("C5 4096 perturbed double-wishbone geometries x 256-step bump sweep (one GPU)",
                           lambda: ensemble_problem(4096, 256), device, 20, 3, roles_geometry=geometry_path("geometry.yaml")),
        ]
        line["dropin"] = measure_dropin(device)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(STEPS_PER_RANK)
    # the figures a reader of the line's first 2000 characters should not miss, right behind the contract's own fields:
    # what the same launch gives without the amortised design-state pass, a fresh program's first launch, the evaluated rates
    if c5_sharded:
        line["rccl"] = c5_sharded.pop("rccl", None)
        line["c5_sharded"] = c5_sharded
    digest = {}
    if "own_first_pass" in line:
        digest["own_first_pass"] = line["own_first_pass"]["value"]
    if "one_shot" in line:
        digest["one_shot_first_launch"] = line["one_shot"]["value_first_launch"]
    if "sustained" in line:
        digest["sustained"] = line["sustained"]["value"]
    for cfg in line.get("other_configs", []):
        tag = cfg["workload"].split(" ")[0].lower()
        if "cold" in cfg:
            digest[f"{tag}_cold"] = cfg["cold"]["value"]
        if isinstance(cfg.get("evaluated"), dict) and "value" in cfg["evaluated"]:
            digest[f"{tag}_evaluated_states_per_s"] = cfg["evaluated"]["value"]
    if digest:
        digest["note"] = "solves/s unless named otherwise; each is detailed under its own key further on"
        head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data")
        line = {**{k: line[k] for k in head if k in line}, "digest": digest, **{k: v for k, v in line.items() if k not in head}}
    # ... and once more as the line's LAST key, for a reader who only sees its tail: headline, roofline, CPU baseline, the
    # sharded config-5 figures and the evaluated rates, in under a screen
    tail = {"value": line["value"], "unit": line["unit"], "n_gpus": world, "ms_per_step": line["ms_per_step"],
            "roofline": {k: line["roofline"].get(k) for k in ("kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic")}}
    if "cpu_baseline" in line:
        tail["cpu_baseline"] = {k: line["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")}
    if isinstance(line.get("c5_sharded"), dict):
        tail["c5_sharded"] = {form: {k: v.get(k) for k in ("value", "solve_only", "exchange_ms", "chunks", "bytes_per_rank", "predicted")}
                              for form, v in line["c5_sharded"].items() if isinstance(v, dict) and "value" in v}
        tail["rccl"] = {k: (line.get("rccl") or {}).get(k) for k in ("world", "backend", "p2p_groups")}
    tail.update({k: v for k, v in digest.items() if k != "note"})
    line["summary"] = tail
    return line


# what a SCALE run should find for BASELINE config 5 (whole-job solves/s; model: measured compute stages
# of one GPU, 60 GB/s per xGMI link and direction, 20 us per grouped call) - printed beside the measurement so that the first
# measured curve can be checked against it
C5_PREDICTED = {"free": {1: 2.4e9, 2: 7.0e8, 4: 1.35e9, 8: 2.48e9}, "metrics": {1: 1.33e9, 2: 1.73e9, 4: 3.04e9, 8: 4.7e9}}


C5_LEG_DEADLINE_S = 240.0


def _with_deadline(fn, seconds, device) -> tuple:
    """``(fn(), False)``; an exception becomes ``{"error": ...}``.  With ``seconds``: run on a helper thread and give up after
    that long - ``({"error": "timed out ..."}, True)`` (the thread is a daemon: the caller is expected to leave the process)."""
    import threading

    box = {}

    def run():
        try:
            torch.cuda.set_device(device)
            box["value"] = fn()
        except Exception as exc:  # noqa: BLE001 - an extra leg must not take the line down
            box["value"] = {"error": f"{type(exc).__name__}: {exc}"}

    if seconds is None:
        run()
        return box["value"], False
    stream = torch.cuda.current_stream(device)
    worker = threading.Thread(target=lambda: (torch.cuda.set_stream(stream), run()), daemon=True)
    worker.start()
    worker.join(seconds)
    if worker.is_alive():
        return {"error": f"timed out after {seconds:.0f} s (the exchange did not complete on this node)"}, True
    return box["value"], False


def measure_c5_sharded(world: int, rank: int, device, steps: int = 5, warmup: int = 2) -> dict:
    """
    BASELINE config 5 (4096 perturbed double-wishbone geometries x 256 steps, "batch sharded across 8 x MI355X") through
    dist.ShardedEnsemble in THIS process group, in the two forms north_star's exchange can take: `free` - every rank ends up
    with every solved state's free coordinates + status byte (145 B per solve over the links, pipelined in chunks) - and
    `metrics` - the evaluated ensemble: every rank solves AND evaluates its shard in one launch per chunk and four metric
    columns + the status byte travel (33 B per state).  Collective: every rank calls it; rank 0 gets the numbers.
    `value` = all 1048576 problems / the slowest rank's step (barrier + synchronize on both sides, like the headline).
    At world == 1 the same keys come from the one-GPU pipeline (nothing travels).
    """
    import torch.distributed as dist

    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.dist import ShardedEnsemble, shard_range
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles
    from open_kinematics_amd.workloads import ensemble_problem, geometry_path

    n_geom, spg = 4096, 256
    program, table, rel = ensemble_problem(n_geom, spg)
    dp = DeviceProgram(program, device)
    dp.enable_evaluation(corner_roles(load_geometry(geometry_path("geometry.yaml")), program))
    table_dev = torch.as_tensor(table, device=device)
    bump = program.n_targets - 1
    columns = [("camber", None), ("camber", bump), ("roadwheel_angle", bump), (21, bump)]  # (21: the wheel centre's z rate)
    glo, ghi = shard_range(n_geom, rank, world)
    out = {}

    def timed(fn) -> float:
        """ms per call, max over ranks: barrier + synchronize on both sides of `steps` calls."""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)
        ms = (time.perf_counter() - t0) / steps * 1e3
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        return ms

    for form, kw in (("free", dict(records=False, info="status")), ("metrics", dict(metric_columns=columns))):
        pipe = ShardedEnsemble(dp, table_dev, rel, spg, chain_len=1, predictor=False, **kw)
        step_ms = timed(pipe.step)
        groups_per_step = 0
        if world > 1:
            before = pipe.p2p_groups
            pipe.step()
            groups_per_step = pipe.p2p_groups - before
        exchange_ms = timed(pipe.exchange_only) if world > 1 else 0.0
        # the compute stage alone: this rank's shard as ONE launch (what `solve_only` means everywhere in this file)
        n_local = (ghi - glo) * spg
        info = torch.empty((n_local, 40), dtype=torch.uint8, device=device)
        skw = dict(info_out=info, chain_len=1, predictor=False, geom_pos=pipe.my_pos, geom_row_param=pipe.my_param, steps_per_geometry=spg)
        if form == "metrics":
            evb = torch.empty((n_local, 1 + program.n_targets, 24), dtype=torch.float64, device=device)
            launch = dp.plan_evaluated(pipe.local_targets, eval_out=evb, output="none", **skw)
        else:
            free = torch.empty((n_local, program.n_free, 3), dtype=torch.float64, device=device)
            launch = dp.plan(pipe.local_targets, out=free, output="free", **skw)
        solve_ms = timed(launch)
        _, ok = info_summary(info)
        out[form] = {"value": pipe.n_total / (step_ms * 1e-3), "unit": "constraint solves/s" + (", every solve evaluated" if form == "metrics" else ""),
                     "step_ms": step_ms, "solve_only": pipe.n_total / (solve_ms * 1e-3), "solve_only_ms_per_rank": solve_ms,
                     "exchange_ms": exchange_ms, "chunks": pipe.chunks, "bytes_per_rank": pipe.exchange_bytes_per_rank,
                     "p2p_groups_per_step": groups_per_step, "all_converged": ok,
                     "predicted": C5_PREDICTED[form].get(world), "predicted_from": "DESIGN.md section 8"}
        del pipe
    backend = dist.get_backend() if world > 1 else None
    out["rccl"] = {"world": world, "backend": backend, "p2p_groups": sum(out[f]["p2p_groups_per_step"] for f in ("free", "metrics")),
                   "note": "grouped point-to-point calls (batch_isend_irecv = one ncclGroup each: every peer over its own link at once) "
                           "issued per step of the two forms; backend nccl = RCCL over xGMI" if world > 1 else
                           "one rank: nothing travels (the same keys as a multi-GPU line, from the one-GPU pipeline)"}
    out["workload"] = ("BASELINE config 5: 4096 perturbed double-wishbone geometries (sigma = 1 mm, seed 0) x 256-step bump sweep = "
                       "1048576 problems, geometry-major shards, strong scaling")
    dp.close()
    return out if rank == 0 else {}


def run_c5(args, world: int, rank: int, device) -> dict:
    """BASELINE config 5: 4096 perturbed geometries x 256 steps, sharded geometry-major over the ranks (strong scaling:
    the total is fixed).  Reports the solve-only and the exchange-inclusive rate separately."""
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.dist import ShardedEnsemble, shard_range
    from open_kinematics_amd.workloads import ensemble_problem

    n_geom, spg = 4096, 256
    program, table, rel = ensemble_problem(n_geom, spg)
    dp = DeviceProgram(program, device)
    glo, ghi = shard_range(n_geom, rank, world)
    spans = [tuple(spg * g for g in shard_range(n_geom, r, world)) for r in range(world)]
    table_dev = torch.as_tensor(table, device=device)
    for _ in range(2):  # the second round is the one reported: the first pays the lazy load of every kernel involved
        torch.cuda.synchronize(device)
        r0 = time.perf_counter()
        gpos_all, gparam_all = dp.rebind(table_dev)   # replicated inputs (1.5 MB): the receiving side's expand needs every geometry
        gpos, gparam = gpos_all[glo:ghi].contiguous(), gparam_all[glo:ghi].contiguous()
        targets = dp.ensemble_targets(gpos, rel)
        torch.cuda.synchronize(device)
        rebind_ms = (time.perf_counter() - r0) * 1e3
    n_local, n_total = (ghi - glo) * spg, n_geom * spg
    # N > 1: the solve writes the free coordinates - the exchange's payload - and every rank expands the gathered block;
    # one rank: the records themselves
    out = torch.empty((n_local, program.n_free if world > 1 else program.n_out, 3), dtype=torch.float64, device=device)
    info = torch.empty((n_local, 40), dtype=torch.uint8, device=device)
    launch = dp.plan(targets, out=out, info_out=info, chain_len=args.chain_len if args.chain_len != -1 else 1, predictor=False,
                     geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=spg, output="free" if world > 1 else "records")
    # N > 1: the pipelined exchange (dist.ShardedEnsemble): the shard in chunks of whole geometries, chunk k + 1 solving while
    # chunk k travels (coordinates + info records, one grouped point-to-point call, from and into their final place) and
    # chunk k - 1 is expanded on a third stream; --c5-gather free leaves the expand out (coordinates on every rank)
    pipe = None
    metric_columns = None
    if args.c5_gather == "metrics":
        from open_kinematics_amd.input import load_geometry
        from open_kinematics_amd.metrics import corner_roles
        from open_kinematics_amd.workloads import geometry_path

        dp.enable_evaluation(corner_roles(load_geometry(geometry_path("geometry.yaml")), program))
        bump = program.n_targets - 1
        metric_columns = [("camber", None), ("camber", bump), ("roadwheel_angle", bump), (21, bump)]  # (21: the wheel centre's z rate)
    if world > 1 or metric_columns:
        pipe = ShardedEnsemble(dp, table_dev, rel, spg, chunks=args.c5_chunks or None, records=args.c5_gather == "records",
                               info=args.c5_info, metric_columns=metric_columns,
                               chain_len=args.chain_len if args.chain_len != -1 else 1, predictor=False)

    def step(k, start, end):
        if start is not None:
            start.record()
        if pipe is not None:
            pipe.step()
        else:
            launch()
        if end is not None:
            end.record()

    elapsed, step_ms = timed_region(step, lambda: None, args.steps, args.warmup, world, device, per_launch_events=True)
    kernel_ms = step_ms
    if pipe is not None:  # the solve alone, for `solve_only`: the whole shard as one launch, outside the timed region
        if metric_columns:  # (the evaluated solve of the shard, no positions written)
            evb = torch.empty((n_local, 1 + program.n_targets, 24), dtype=torch.float64, device=device)
            launch = dp.plan_evaluated(targets, info_out=info, eval_out=evb, output="none", chain_len=1, predictor=False,
                                       geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=spg)
        _, kernel_ms = time_launches(launch, max(3, min(args.steps, 10)), 2, device)
        info = pipe.info_local if pipe.status_only else pipe.info_full[glo * spg : ghi * spg]
    nfev_mean, ok = info_summary(info)
    if rank != 0:
        return {}
    bytes_per = algorithmic_bytes_per_solve(program, spg)
    if metric_columns:  # targets in, the info record and the [1 + T][24] evaluation rows out, no positions
        bytes_per = 8 * program.n_targets + 40 + 8 * 24 * (1 + program.n_targets)
    gbs = bytes_per * n_local / (kernel_ms * 1e-3) / 1e9
    free_bytes = n_local * program.n_free * 24
    return {
        "metric": "constraint solves/sec (sweep steps/sec), double-wishbone sensitivity ensemble" + (", every solve evaluated (tangents, 19 metrics, their derivatives)" if metric_columns else ""),
        "value": n_total * args.steps / elapsed,
        "unit": "constraint solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: 4096 perturbed double-wishbone geometries (sigma = 1 mm, seed 0) x 256-step bump "
                               "sweep = 1048576 problems, geometry-major shards (a rank owns whole geometries)",
                   "geometries_per_gpu": ghi - glo, "problems_per_gpu": n_local, "kernel": dp.kernel,
                   "start": "independent cold starts from each geometry's design state (SURVEY.md section 8d)" if args.chain_len in (-1, 1)
                            else f"chains of {args.chain_len}",
                   "lm_evaluations_mean": nfev_mean, "all_converged": ok, "rebind_ms": rebind_ms,
                   "exchange": ("evaluated ensemble: every rank solves AND evaluates its shard (one launch per chunk, no positions "
                                "written); %d metric columns of every state (8 B each) + a status byte travel, %d chunk(s)"
                                % (len(metric_columns), pipe.chunks)) if metric_columns else
                               ("pipelined: the shard in %d chunks of whole geometries, chunk k + 1 solving while chunk k travels "
                                "(coordinates + info records in one grouped point-to-point call, from and into their final place)%s"
                                % (pipe.chunks, " and chunk k - 1 is expanded into records on a third stream" if pipe.records else
                                   "; coordinates only, no expand (--c5-gather free)")) if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                     "kernel": ("okx_lane_evsolve_g" if metric_columns and (dp.evaluation & 2) else
                                "okx_lane_solve_g" if 0 < dp.lane_threshold <= n_local and (dp.lane_bodies & 1) else
                                "okx_quad_solve_g" if dp.kernel == "quad" else "okx_solve_kernel"),
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_solve": bytes_per},
        "solve_only": {"value": n_total / (kernel_ms * 1e-3) if world > 1 else n_local / (kernel_ms * 1e-3),
                       "kernel_ms_max_over_ranks": kernel_ms},
        "exchange": {"bytes_sent_per_rank_per_step": pipe.exchange_bytes_per_rank if pipe is not None else 0,
                     "bytes_received_per_rank_per_step": (n_total - n_local) * ((8 * len(metric_columns) if metric_columns else program.n_free * 24)
                                                                                + (1 if args.c5_info == "status" or metric_columns else 40)) if world > 1 else 0,
                     "chunks": pipe.chunks if pipe is not None else 0, "step_ms_with_exchange": step_ms if world > 1 else None},
    }


if __name__ == "__main__":
    main()
