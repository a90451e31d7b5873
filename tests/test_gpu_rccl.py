"""
What a one-GPU box can exercise of the N > 1 bench path on REAL RCCL: a world-size-1 `nccl` process group, the
asynchronous `all_gather_into_tensor` of the solved free coordinates on RCCL's stream, the pipeline's stream ordering and
the expand of the gathered block (bench.py --rccl-world-one).  The N > 1 logic itself is covered by the two-rank gloo
tests in tests/test_dist.py.
"""

import json
import os
import subprocess
import sys

import pytest

from conftest import REPO, gpu_available

pytestmark = pytest.mark.gpu


def test_bench_step_with_an_rccl_all_gather_on_one_rank():
    if not gpu_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
                           "--rccl-world-one", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    line = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    assert line["n_gpus"] == 1 and line["config"]["all_converged"]
    assert line["exchange"]["collective"] == "all_gather_into_tensor (RCCL)"
    assert line["solve_only"]["value"] >= line["value"] > 0.0   # the exchange-inclusive rate cannot beat the solve alone
    assert line["roofline"]["kernel_ms"] > 0.0


def test_two_rank_bench_rehearsal_on_one_gpu():
    """The N = 2 bench step end to end on a one-GPU box: two ranks (both on cuda:0), index shards, the solve writing its
    free coordinates into the send buffer, the pipelined all-gather (over gloo here), one expand per step of the gathered
    block, a fixed-count preheat (every step is a collective call), MAX over ranks.  The numbers mean nothing; the line's
    structure and the agreement of every rank on the step count do."""
    if not gpu_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "10",
                           "--warmup", "3", "--preheat-ms", "2", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True,
                          text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    line = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    assert line["n_gpus"] == 2 and line["config"]["all_converged"] and line["scaling"] == "weak"
    assert line["config"]["problems_per_gpu"] == 16384 and line["preheat"]["steps"] == 100
    assert line["exchange"]["bytes_sent_per_rank_per_step"] == 16384 * 6 * 24      # the free coordinates, not the records
    assert line["exchange"]["bytes_received_per_rank_per_step"] == 16384 * 6 * 24
    assert line["solve_only"]["value"] >= line["value"] > 0.0
    # the default line carries BASELINE config 5 through the sharded pipeline of the SAME process group, in both forms
    assert line["rccl"]["world"] == 2 and line["rccl"]["backend"] == "gloo" and line["rccl"]["p2p_groups"] >= 2
    for form, per_state in (("free", 6 * 24 + 1), ("metrics", 4 * 8 + 1)):
        leg = line["c5_sharded"][form]
        assert leg["all_converged"] and leg["value"] > 0.0 and leg["solve_only"] >= leg["value"]
        assert leg["bytes_per_rank"] == 2048 * 256 * per_state and leg["chunks"] >= 1 and leg["exchange_ms"] > 0.0
        assert leg["p2p_groups_per_step"] == leg["chunks"] and leg["predicted"] is not None
    assert line["summary"]["c5_sharded"]["free"]["value"] == line["c5_sharded"]["free"]["value"]


def test_two_rank_c5_pipeline_rehearsal_on_one_gpu():
    """BASELINE config 5's N = 2 step on a one-GPU box (both ranks on cuda:0, gloo in place of RCCL): geometry-major shards
    cut into chunks, every chunk's solve writing coordinates and info records into their final place, the grouped
    point-to-point exchange, the expand of the pieces on a third stream - the same with coordinates only, and the
    evaluated ensemble (every rank evaluates its shard, metric columns travel)."""
    if not gpu_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    for extra in ([], ["--c5-gather", "free", "--c5-chunks", "5"], ["--c5-gather", "metrics"]):
        proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", "c5", "--gpus", "2", "--rehearse-on-one-gpu",
                               "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"] + extra, env=env,
                              capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
        line = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")][-1]
        assert line["n_gpus"] == 2 and line["config"]["all_converged"] and line["scaling"] == "strong"
        assert line["config"]["problems_per_gpu"] == 2048 * 256
        if "metrics" in extra:  # the evaluated ensemble: four metric columns and the status byte per state
            assert line["exchange"]["chunks"] == 2 and line["exchange"]["bytes_sent_per_rank_per_step"] == 2048 * 256 * (4 * 8 + 1)
            assert "evaluated" in line["metric"] and line["roofline"]["kernel"] == "okx_lane_evsolve_g"
            continue
        assert line["exchange"]["chunks"] == (5 if extra else 8)
        assert line["exchange"]["bytes_sent_per_rank_per_step"] == 2048 * 256 * (6 * 24 + 1)   # coordinates + one status byte
        assert line["solve_only"]["value"] > 0.0 and line["value"] > 0.0


def test_one_gpu_bench_line_keeps_the_contract():
    """`python bench.py --steps 20 --warmup 5` as the driver runs it: one JSON line with the contract's fields, the roofline
    and cpu_baseline objects, the K timed steps as one HIP graph submitted through the HIP runtime directly - and the same
    through torch's own calls (--torch-submit)."""
    if not gpu_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    for extra in ([], ["--torch-submit", "--no-cpu-baseline"]):
        proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras"] + extra,
                              env=env, capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        line = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline"):
            assert key in line, key
        assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["dtype"] == "f64" and line["vs_baseline"] is None
        assert line["config"]["all_converged"] and "workload" in line["config"]
        roof = line["roofline"]
        assert roof["bound"] == "hbm" and roof["kernel"] == "okx_quad_cold_u" and 0.0 < roof["frac"] < 1.0
        assert abs(roof["achieved"] - 392.0 * 16384 / (roof["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * roof["achieved"]
        assert roof["kernel_ms"] <= line["ms_per_step"]                       # the kernel cannot take longer than the step that holds it
        assert abs(line["value"] - 16384 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
        sub = line["submission"]
        assert sub["mode"] == "hip graph"
        if extra:
            assert sub["calls"].startswith("torch")
        else:
            assert sub["calls"].startswith("hipEventRecord") and set(sub["host_us"]) == {"submit", "poll_until_done", "drain_and_synchronize"}
            base = line["cpu_baseline"]
            assert base["kind"] == "port" and base["cores"] >= 1 and base["value"] > 0.0
            # the one-GPU line has the multi-GPU line's keys (from the one-GPU pipeline: nothing travels), and ends in a summary
            assert line["rccl"] == {**line["rccl"], "world": 1, "backend": None, "p2p_groups": 0}
            for form in ("free", "metrics"):
                leg = line["c5_sharded"][form]
                assert leg["all_converged"] and leg["value"] > 1e8 and leg["bytes_per_rank"] == 0 and leg["exchange_ms"] == 0.0
            assert list(line)[-1] == "summary" and line["summary"]["roofline"]["frac"] == roof["frac"]
            assert line["summary"]["cpu_baseline"]["value"] == base["value"]
