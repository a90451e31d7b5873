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
