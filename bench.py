#!/usr/bin/env python3
"""
Headline benchmark: constraint solves/sec on the double-wishbone bump sweep (BASELINE.json).

One "step" = one pass of the hot path over one batch: every rank solves a 16384-step fp64
bump sweep (BASELINE config 2; inputs already resident in HBM) with ONE launch of the solve
kernel (the runtime-specialised quad kernel `okx_quad_solve_u`); with N > 1 ranks
the global sweep (N x 16384 steps) is sharded by index and the solved positions are all-gathered
over RCCL inside the timed step.  Prints one JSON line on rank 0.

  python bench.py --gpus 1 --steps 200 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

STEPS_PER_RANK = 16384
CHAIN_LEN = -1                # auto: one chain per resident problem slot (16384 steps fit the chip: all cold starts)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # vendor fp64 vector peak, secondary ceiling


def algorithmic_bytes_per_solve(program) -> int:
    """8*T targets in + 24*P_out positions out + 16 B info (we write 40)."""
    return 8 * program.n_targets + 24 * program.n_out + 16


def estimated_flops_per_evaluation(program, stats) -> float:
    """Analytic fp64 flop count of one LM iteration: rows + J^T J + Cholesky + solves."""
    n, m = program.n_vars, program.n_residuals
    rows = 60.0 * m
    normal = 2.0 * 9.0 * stats["contrib"] + 2.0 * 3.0 * stats["contrib"]
    chol = n ** 3 / 3.0 + 2.0 * n * n
    return rows + normal + chol


PROFILE_DIR = os.path.join("profiles", "r01")
TRAFFIC_SUMMARY = "profiles/r01/bench_c2_pred_pmc_traffic.json"


def measured_traffic() -> tuple:
    """
    HBM bytes per launch of the solve kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of this same command, see
    profiles/r01/bench_c2_quad_pmc_traffic.json for command, units and corrections).  bench.py
    cannot run the profiler on itself, so the figure is read back from that summary.
    """
    path = os.path.join(REPO, TRAFFIC_SUMMARY)
    try:
        with open(path, "r", encoding="utf-8") as fh:
            summary = json.load(fh)
        return float(summary["traffic_bytes_per_launch"]), os.path.relpath(path, REPO)
    except (OSError, KeyError, ValueError):
        return None, None


_SHARD_CODE = """
import json, sys, time
sys.path.insert(0, {repo!r})
from open_kinematics_amd.workloads import bump_sweep_problem
from oracle.oracle import Oracle
lo, hi, n = {lo}, {hi}, {n}
program, targets = bump_sweep_problem(n, line_mode="softnorm")
orc = Oracle(program)
orc.sweep(targets[lo:lo + 32])
t0 = time.perf_counter()
res = orc.sweep(targets[lo:hi])
print(json.dumps([time.perf_counter() - t0, int(res.first_failed_step), float(res.info["nfev"].sum())]))
"""


def _cpu_chain(lo: int, hi: int, n_steps: int):
    """One contiguous shard of the sweep as a sequential warm-started chain, in this process."""
    from open_kinematics_amd.workloads import bump_sweep_problem
    from oracle.oracle import Oracle

    program, targets = bump_sweep_problem(n_steps, line_mode="softnorm")
    orc = Oracle(program)
    orc.sweep(targets[lo:lo + 32])  # warm the library / caches
    t0 = time.perf_counter()
    res = orc.sweep(targets[lo:hi])
    return [time.perf_counter() - t0, int(res.first_failed_step), float(res.info["nfev"].sum())]


def cpu_baseline(n_steps: int) -> dict:
    """
    Oracle = reference-shaped CPU path (sequential warm-started MINPACK LM, default tolerances) on
    the GPU box's host cores: the sweep is cut into one contiguous chain per core (SURVEY.md §8d
    (ii)).  Workers are plain child interpreters (subprocess, hard timeout) that never touch the
    GPU; if they cannot be run the figure falls back to one core in this process.
    """
    import subprocess

    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    results = None
    if cores > 1:
        bounds = [(n_steps * k // cores, n_steps * (k + 1) // cores) for k in range(cores)]
        env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, "-c", _SHARD_CODE.format(repo=REPO, lo=lo, hi=hi, n=n_steps)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
                 for lo, hi in bounds]
        try:
            results = []
            for proc in procs:
                out, _ = proc.communicate(timeout=180)
                results.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001 - any worker trouble: measure one core here instead
            results = None
            for proc in procs:
                if proc.poll() is None:
                    proc.kill()
    if results is None:
        cores = 1
        results = [_cpu_chain(0, n_steps, n_steps)]
    if any(r[1] != -1 for r in results):
        raise RuntimeError("cpu baseline: oracle sweep failed")
    slowest = max(r[0] for r in results)
    nfev = sum(r[2] for r in results) / n_steps
    return {
        "value": n_steps / slowest,
        "unit": "constraint solves/s",
        "cores": cores,
        "kind": "port",
        "sample": f"full {n_steps}-step bump sweep as {cores} contiguous warm-started chain(s), one process per host "
                  f"core, MINPACK LM ftol=1e-5 xtol=gtol=1e-9 (reference defaults), slowest chain {slowest:.2f} s, "
                  f"mean nfev {nfev:.1f}; per core {n_steps / sum(r[0] for r in results):.0f} solves/s",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-len", type=int, default=None, help="override: 1 = independent cold starts")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank uses cuda:0 and the exchange runs over gloo "
                         "(exercises the sharding / pipeline / rebuild logic, not RCCL; the numbers mean nothing)")
    ap.add_argument("--no-predictor", action="store_true",
                    help="chain heads start from the design state instead of the fitted polynomial model")
    args = ap.parse_args()

    global CHAIN_LEN
    if args.chain_len is not None:
        CHAIN_LEN = args.chain_len
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the solve path)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from open_kinematics_amd import _lib
    from open_kinematics_amd._abi import HostProgram
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.dist import FreeGatherPipeline, shard_range
    from open_kinematics_amd.workloads import bump_sweep_problem

    n_total = STEPS_PER_RANK * world
    program, targets_all = bump_sweep_problem(n_total)
    lo, hi = shard_range(n_total, rank, world)
    dp = DeviceProgram(program, device)
    targets = torch.as_tensor(targets_all[lo:hi], device=device).contiguous()
    info = torch.empty((hi - lo, 40), dtype=torch.uint8, device=device)
    # Two output slots: with N > 1 ranks the all-gather of step k (RCCL stream) overlaps the solve
    # of step k + 1 (launch stream), see dist.GatherPipeline.  The exchange ships the free coordinates
    # of each solve (144 B) and every rank rebuilds the full positions (360 B) itself
    # (dist.FreeGatherPipeline / okx_expand_positions_batch).  One rank: slot 0 only, no exchange.
    pipe = FreeGatherPipeline(hi - lo, program.n_out, dp.free_out_index, dp.expand, torch.float64, device)
    # pre-bound launches: per step the host only makes the C-ABI call (the kernel is ~40 us long)
    # Chain-head predictor: fitted once per program over this rank's target box (8 node solves for the
    # one varying target; setup, like the kernel compile) — every cold start of the timed launches then
    # begins at the fitted polynomial instead of the design state.  Same solutions to step_tol.
    use_predictor = not args.no_predictor and dp.fit_predictor(targets)
    launches = [dp.plan(targets, out=buf, info_out=info, chain_len=CHAIN_LEN, predictor=use_predictor) for buf in pipe.local]

    def step(k: int):
        pipe.begin(k)
        launches[k % len(launches)]()
        pipe.submit(k)

    for k in range(args.warmup):
        step(k)
    pipe.drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)

    # Kernel duration from HIP events on the launch stream (torch's current stream IS the stream
    # the C-ABI call launches on).  One rank: a single event pair brackets the K back-to-back
    # launches of the timed region (average = elapsed / K; per-launch pairs would add two stream
    # markers per 40 us kernel to the region being timed).  Several ranks: per-launch pairs, so the
    # all-gather between launches is excluded.
    n_pairs = args.steps if world > 1 else 1
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_pairs)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_pairs)]
    t0 = time.perf_counter()
    if world == 1:
        launch = launches[0]
        starts[0].record()
        for k in range(args.steps):
            launch()
        ends[0].record()
    else:
        for k in range(args.steps):
            pipe.begin(k)
            starts[k].record()
            launches[k % len(launches)]()
            ends[k].record()
            pipe.submit(k)
        pipe.drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # average launch duration over the timed region (one rank: includes the ~1 us launch-to-launch
    # gaps of the back-to-back launches; agrees with the rocprofv3 kernel trace to < 1 %)
    kernel_ms = float(np.sum([s.elapsed_time(e) for s, e in zip(starts, ends)])) / args.steps
    host_info = info.cpu().numpy().view(np.dtype([("max_residual", "<f8"), ("cost", "<f8"), ("last_step", "<f8"),
                                                  ("iterations", "<i4"), ("nfev", "<i4"), ("flags", "<i4"),
                                                  ("reserved", "<i4")])).reshape(-1)
    ok = bool(np.all((host_info["flags"] & 7) == 1))

    if rank == 0:
        import ctypes as C

        stats_raw = (C.c_int32 * 8)()
        _lib.load().okx_plan_stats(HostProgram(program).byref(), stats_raw)
        stats = dict(zip(["n", "m", "pairs", "contrib", "active", "js_stride", "lda", "lds_bytes"], list(stats_raw)))
        bytes_per_solve = algorithmic_bytes_per_solve(program)
        traffic, traffic_src = measured_traffic() if world == 1 and CHAIN_LEN == -1 and dp.kernel == "quad" else (None, None)
        achieved_gbs = bytes_per_solve * (hi - lo) / (kernel_ms * 1e-3) / 1e9
        nfev_mean = float(host_info["nfev"].mean())
        flops = estimated_flops_per_evaluation(program, stats) * nfev_mean * (hi - lo)
        line = {
            "metric": "constraint solves/sec (sweep steps/sec), double-wishbone bump sweep",
            "value": n_total * args.steps / elapsed,
            "unit": "constraint solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
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
                "start": ("chain_len=-1 (auto): one chain per resident problem slot; 16384 steps fit the "
                          "chip's slots, so every step is an independent solve (SURVEY.md §8d); longer sweeps "
                          "become warm-started chains (solver.py:774). " +
                          ("Each solve starts from the degree-7 Chebyshev model of the solution over the sweep's "
                           "target box, fitted once per program from 8 node solves (okx_program_fit_predictor, "
                           "setup); the LM iteration and its stopping rules are unchanged (--no-predictor: "
                           "cold starts from the design state)" if use_predictor else
                           "Each solve is a cold start from the design state"))
                         if dp.kernel == "quad" else
                         "sweep split into contiguous chunks, one per resident wavefront; chunk head cold "
                         "(design state), later steps warm-started from their predecessor "
                         "(reference semantics, solver.py:774)",
                "lm_evaluations_mean": nfev_mean,
                "predictor": bool(use_predictor),
                "all_converged": ok,
                "exchange": ("RCCL all-gather of the solved free coordinates every step (every rank rebuilds all positions "
                             "from them), overlapped with the next step's solve (two output slots); %d B per rank per step "
                             "instead of %d B of positions, i.e. every rank receives %d B per step: at the single-GPU solve "
                             "rate that is still more than xGMI can deliver, so N > 1 runs at the exchange's rate "
                             "(DESIGN.md section 8)" % (STEPS_PER_RANK * program.n_free * 24, STEPS_PER_RANK * program.n_out * 24,
                                                    (world - 1) * STEPS_PER_RANK * program.n_free * 24))
                            if world > 1 else "none",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "okx_quad_solve_u" if dp.kernel == "quad" else "okx_solve_kernel",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_solve": bytes_per_solve,
                "note": "path is latency/fp64-issue bound, not HBM bound (SURVEY.md §7 H4): see compute",
            },
            "compute": {
                "unit": "TFLOP/s",
                "achieved": flops / (kernel_ms * 1e-3) / 1e12,
                "peak": FP64_VECTOR_PEAK_TFLOPS,
                "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "flops_per_solve_estimate": flops / (hi - lo),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(STEPS_PER_RANK)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
