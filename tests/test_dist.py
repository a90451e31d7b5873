"""N > 1 path on CPU: world_size-2 gloo run of the index sharding + all-gather exchange."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO
from open_kinematics_amd.dist import FreeGatherPipeline, GatherPipeline, all_gather_rows, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 16, 16385, 1048576):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _fake_solve(targets: torch.Tensor) -> torch.Tensor:
    """Stand-in for the device solve in the CPU test: a deterministic function of the targets."""
    t = targets[:, :1]
    return torch.stack([t * 1.0, t * 2.0 + 1.0, t * t], dim=2).expand(-1, 5, -1).contiguous()


def _worker(rank: int, world: int, port: int, n_total: int, out_dir: str) -> None:
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    targets = torch.linspace(-60.0, 80.0, n_total, dtype=torch.float64).reshape(-1, 1)
    lo, hi = shard_range(n_total, rank, world)
    local = _fake_solve(targets[lo:hi])
    full = all_gather_rows(local, n_total)
    torch.save(full, os.path.join(out_dir, f"rank{rank}.pt"))
    if n_total % world == 0:
        # the bench's pipelined exchange: gather of step k overlaps the "solve" of step k + 1
        pipe = GatherPipeline(hi - lo, (5, 3), torch.float64, "cpu")
        for k in range(5):
            out = pipe.begin(k)
            out.copy_(_fake_solve(targets[lo:hi] + float(k)))
            pipe.submit(k)
        torch.save(pipe.drain().clone(), os.path.join(out_dir, f"pipe{rank}.pt"))
        # the compact exchange: "free" points 1 and 3 travel, the receiver rebuilds the other three of five
        index = torch.tensor([1, 3])

        def expand(free, out):  # stand-in for DeviceProgram.expand: _fake_solve's points are all equal
            out.copy_(free[:, :1, :].expand(-1, 5, -1))

        compact = FreeGatherPipeline(hi - lo, 5, index, expand, torch.float64, "cpu")
        for k in range(5):
            out = compact.begin(k)
            out.copy_(_fake_solve(targets[lo:hi] + float(k)))
            compact.submit(k)
        torch.save(compact.drain().clone(), os.path.join(out_dir, f"compact{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 101])
def test_two_rank_gloo_gather_reassembles_the_sweep(tmp_path, n_total):
    port = 29500 + (os.getpid() + n_total) % 2000
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    expect = _fake_solve(torch.linspace(-60.0, 80.0, n_total, dtype=torch.float64).reshape(-1, 1))
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert got.shape == expect.shape and torch.equal(got, expect)
        if n_total % 2 == 0:
            last = _fake_solve(torch.linspace(-60.0, 80.0, n_total, dtype=torch.float64).reshape(-1, 1) + 4.0)
            assert torch.equal(torch.load(os.path.join(tmp_path, f"pipe{rank}.pt")), last)
            assert torch.equal(torch.load(os.path.join(tmp_path, f"compact{rank}.pt")), last)


def test_single_process_pipeline_degenerates_to_the_local_buffer():
    pipe = GatherPipeline(4, (3,), torch.float64, "cpu")
    for k in range(3):
        pipe.begin(k).fill_(float(k))
        pipe.submit(k)
    assert torch.equal(pipe.drain(), torch.full((4, 3), 2.0, dtype=torch.float64))


def test_single_process_compact_pipeline_degenerates_to_the_local_buffer():
    pipe = FreeGatherPipeline(4, 3, torch.tensor([0, 2]), lambda free, out: None, torch.float64, "cpu")
    for k in range(3):
        pipe.begin(k).fill_(float(k))
        pipe.submit(k)
    assert torch.equal(pipe.drain(), torch.full((4, 3, 3), 2.0, dtype=torch.float64))


def test_single_process_gather_is_identity():
    x = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    assert all_gather_rows(x, 4) is x
    with pytest.raises(ValueError):
        all_gather_rows(x, 5)
