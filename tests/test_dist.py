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
        # the compact exchange: the solve writes the "free" points 1 and 3 (output = free) into the send buffer, they
        # travel, and the receiver rebuilds all five points of every rank's rows
        def expand(free, out):  # stand-in for DeviceProgram.expand: _fake_solve's points are all equal
            out.copy_(free[:, :1, :].expand(-1, 5, -1))

        compact = FreeGatherPipeline(hi - lo, 5, 2, expand, torch.float64, "cpu")
        assert compact.output == "free" and compact.solve_buffers[0].shape == (hi - lo, 2, 3)
        assert compact.bytes_sent_per_step == (hi - lo) * 2 * 24
        for k in range(5):
            out = compact.begin(k)
            out.copy_(_fake_solve(targets[lo:hi] + float(k))[:, [1, 3]])
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


class _StandInProgram:
    """
    CPU stand-in for ``DeviceProgram`` with the same ensemble surface (``rebind``, ``ensemble_targets``, ``solve``
    with geometry tables, ``free_out_index``, ``expand``): 5 output points, point 0 fixed (from the geometry),
    points 1 and 3 "free" (a function of target and geometry), points 2 and 4 "derived" (midpoint / offset).
    """

    class program:  # noqa: N801 - attribute bag like ConstraintProgram
        n_out, n_free, n_targets = 5, 2, 1

    free_out_index = torch.tensor([1, 3])

    def __init__(self, fill_buffers: bool = True):
        self.rebound = []
        self.outputs = []
        self.launch_rows = []
        self.fill_buffers = fill_buffers  # False: fresh tensors come back, as from a program that ignores out= / info_out=

    def rebind(self, table):
        table = torch.as_tensor(table, dtype=torch.float64)
        self.rebound.append(table.shape[0])
        return table.clone(), table[:, :1, :].expand(-1, 4, -1).reshape(table.shape[0], 4, 3).clone()

    def ensemble_targets(self, gpos, relative):
        base = gpos[:, 1, 2:3]                                               # [G, 1]
        return (base[:, None, :] + torch.as_tensor(relative)[None]).reshape(-1, 1)

    @staticmethod
    def _assemble(free, fixed):
        out = torch.empty((free.shape[0], 5, 3), dtype=torch.float64)
        out[:, 0] = fixed
        out[:, 1] = free[:, 0]
        out[:, 3] = free[:, 1]
        out[:, 2] = 0.5 * (free[:, 0] + free[:, 1])
        out[:, 4] = free[:, 1] + fixed
        return out

    def solve(self, targets, *, geom_pos, geom_row_param, steps_per_geometry, output="records", out=None, info_out=None, **kw):
        assert geom_pos.shape[0] * steps_per_geometry == targets.shape[0] and geom_row_param.shape[0] == geom_pos.shape[0]
        fixed = geom_pos[:, 0].repeat_interleave(steps_per_geometry, dim=0)
        t = targets[:, :1]
        solved = torch.stack([t * fixed, t + 2.0 * fixed], dim=1)
        self.outputs.append(output)
        self.launch_rows.append(targets.shape[0])
        info = (targets[:, :1].abs() * 7).to(torch.uint8).expand(-1, 40).contiguous()
        if self.fill_buffers and out is not None:  # like DeviceProgram: the caller's buffers are what is written
            out.copy_(solved if output == "free" else self._assemble(solved, fixed))
            solved = out if output == "free" else solved
        if self.fill_buffers and info_out is not None:
            info_out.copy_(info)
            info = info_out

        class _Result:  # like BatchResult: records with output="records", the free points alone with "free"
            positions = self._assemble(solved, fixed) if output == "records" else None
            free = solved if output == "free" else None
            info_raw = info

        return _Result

    def solve_evaluated(self, targets, *, geom_pos, geom_row_param, steps_per_geometry, output="none", info_out=None, eval_out=None, **kw):
        """Evaluation rows [B, 1 + T, 24] of the stand-in: entry (r, c) = (1 + r) * target + c + the geometry's fixed x."""
        assert output == "none"
        res = self.solve(targets, geom_pos=geom_pos, geom_row_param=geom_row_param, steps_per_geometry=steps_per_geometry,
                         output="free", info_out=info_out)
        fixed = geom_pos[:, 0, 0].repeat_interleave(steps_per_geometry)
        rows = (1.0 + torch.arange(2, dtype=torch.float64))[None, :, None] * targets[:, :1, None] \
            + torch.arange(24, dtype=torch.float64)[None, None, :] + fixed[:, None, None]
        if self.fill_buffers and eval_out is not None:
            eval_out.copy_(rows)
            rows = eval_out

        class _Result:
            eval = rows
            info_raw = res.info_raw

        return _Result

    def expand(self, free, out=None, geom_pos=None, steps_per_geometry=0):
        full = self._assemble(free, geom_pos[:, 0].repeat_interleave(steps_per_geometry, dim=0))
        if out is not None and self.fill_buffers:
            out.copy_(full)
            return out
        return full


def _ensemble_inputs(n_geom: int, steps: int):
    gen = torch.Generator().manual_seed(5)
    table = torch.randn((n_geom, 3, 3), dtype=torch.float64, generator=gen)
    relative = torch.linspace(-6.0, 8.0, steps, dtype=torch.float64).reshape(-1, 1)
    return table, relative


def _ensemble_worker(rank: int, world: int, port: int, n_geom: int, steps: int, out_dir: str) -> None:
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from open_kinematics_amd.dist import solve_sharded

    table, relative = _ensemble_inputs(n_geom, steps)
    for exchange in ("free", "positions"):
        dp = _StandInProgram()
        positions, shard = solve_sharded(dp, relative, hardpoints=table, steps_per_geometry=steps, exchange=exchange, chunks=1)
        torch.save({"positions": positions, "info": shard.info_full, "range": shard.geometry_range,
                    "rebound": dp.rebound, "sent": shard.exchange_bytes_per_rank, "outputs": dp.outputs},
                   os.path.join(out_dir, f"{exchange}{rank}.pt"))
    # the pipelined exchange: the shard in chunks of whole geometries (more chunks than a rank has geometries included),
    # with a program that fills the buffers it is given and with one that returns fresh tensors; records or coordinates only
    for tag, chunks, fill, gather in (("chunk3", 3, True, True), ("chunk8", 8, False, True), ("coords", 2, True, "free")):
        dp = _StandInProgram(fill_buffers=fill)
        got, shard = solve_sharded(dp, relative, gather=gather, hardpoints=table, steps_per_geometry=steps, chunks=chunks)
        if tag == "coords":  # ... and with one status byte per solve beside the coordinates instead of the 40-byte record
            lean, lean_shard = solve_sharded(_StandInProgram(), relative, gather="free", hardpoints=table, steps_per_geometry=steps,
                                             chunks=chunks, info="status")
            assert torch.equal(lean, got) and lean_shard.info_full is None
            assert torch.equal(lean_shard.status_full, shard.info_full[:, 32])
            lo, hi = shard.geometry_range
            assert torch.equal(lean_shard.local.info_raw, shard.info_full[lo * steps : hi * steps])
            assert lean_shard.exchange_bytes_per_rank == (hi - lo) * steps * (2 * 24 + 1)
        torch.save({"result": got, "info": shard.info_full, "free": shard.free_full, "range": shard.geometry_range,
                    "rebound": dp.rebound, "sent": shard.exchange_bytes_per_rank, "launch_rows": dp.launch_rows,
                    "local_free": shard.local.free, "outputs": dp.outputs}, os.path.join(out_dir, f"{tag}{rank}.pt"))
    # the EVALUATED ensemble: every rank evaluates its shard, chosen metric columns travel (8 B each) beside the status byte
    from open_kinematics_amd.dist import ShardedEnsemble

    for fill in (True, False):
        dp = _StandInProgram(fill_buffers=fill)
        pipe = ShardedEnsemble(dp, table, relative, steps, chunks=2, metric_columns=[(3, None), (0, 0), (23, 0)])
        table_k = pipe.step().clone()
        groups = pipe.p2p_groups
        pipe.exchange_only()  # (the exchange stage alone: the same grouped calls once more, nothing solved)
        torch.save({"metrics": table_k, "status": pipe.status_full.clone(), "eval_local": pipe.eval_local.clone(),
                    "sent": pipe.exchange_bytes_per_rank, "range": pipe.geometry_range, "groups": (groups, pipe.p2p_groups),
                    "same_after_exchange_only": torch.equal(pipe.metric_full, table_k)}, os.path.join(out_dir, f"metrics{int(fill)}_{rank}.pt"))
    # the default chunk count comes from the SMALLEST shard: the same on every rank, uneven shards included
    auto = ShardedEnsemble(_StandInProgram(), table, relative, steps, records=False, info="status")
    auto_free = auto.step().clone()
    torch.save({"chunks": auto.chunks, "free": auto_free, "groups": auto.p2p_groups}, os.path.join(out_dir, f"auto{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_geom", [6, 7])
def test_two_rank_gloo_ensemble_is_geometry_major(tmp_path, n_geom):
    """C5's multi-GPU path: geometry-major shards (uneven counts included), local rebind, solve, gather."""
    steps = 4
    port = 31500 + (os.getpid() + n_geom) % 2000
    mp.spawn(_ensemble_worker, args=(2, port, n_geom, steps, str(tmp_path)), nprocs=2, join=True)
    table, relative = _ensemble_inputs(n_geom, steps)
    ref = _StandInProgram()
    gpos, gparam = ref.rebind(table)
    targets = ref.ensemble_targets(gpos, relative)
    expect = ref.solve(targets, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=steps)
    spans = [shard_range(n_geom, r, 2) for r in range(2)]
    for exchange in ("free", "positions"):
        for rank in range(2):
            got = torch.load(os.path.join(tmp_path, f"{exchange}{rank}.pt"))
            assert tuple(got["range"]) == spans[rank]
            assert torch.equal(got["positions"], expect.positions)
            assert torch.equal(got["info"], expect.info_raw)
            rows = (spans[rank][1] - spans[rank][0]) * steps
            payload = rows * (2 if exchange == "free" else 5) * 24 + rows * 40
            assert got["sent"] == payload
            # the compact exchange rebinds the replicated table in full, the positions exchange only its slice
            assert got["rebound"] == [n_geom if exchange == "free" else spans[rank][1] - spans[rank][0]]
            # ... and its solve writes the payload itself (okx_solve_opts.output = free): no packing pass after the solve
            assert got["outputs"] == ["free" if exchange == "free" else "records"]
    # chunked == unchunked, bit for bit, whatever the chunk count and whether or not the program fills the given buffers
    free_expect = expect.positions[:, [1, 3]]
    for rank in range(2):
        lo, hi = spans[rank]
        for tag, chunks in (("chunk3", 3), ("chunk8", 8)):
            got = torch.load(os.path.join(tmp_path, f"{tag}{rank}.pt"))
            assert torch.equal(got["result"], expect.positions) and torch.equal(got["info"], expect.info_raw)
            assert torch.equal(got["free"], free_expect) and torch.equal(got["local_free"], free_expect[lo * steps : hi * steps])
            assert got["rebound"] == [n_geom] and set(got["outputs"]) == {"free"}
            assert got["sent"] == (hi - lo) * steps * (2 * 24 + 40)
            assert sum(got["launch_rows"]) == (hi - lo) * steps and len(got["launch_rows"]) == min(chunks, hi - lo)
        got = torch.load(os.path.join(tmp_path, f"coords{rank}.pt"))  # gather="free": the coordinates, no expand, own slice rebound only
        assert torch.equal(got["result"], free_expect) and torch.equal(got["info"], expect.info_raw)
        assert got["rebound"] == [hi - lo]
    autos = [torch.load(os.path.join(tmp_path, f"auto{rank}.pt")) for rank in range(2)]
    assert autos[0]["chunks"] == autos[1]["chunks"] >= 1 and autos[0]["groups"] == autos[1]["groups"] == autos[0]["chunks"]
    assert torch.equal(autos[0]["free"], free_expect) and torch.equal(autos[1]["free"], free_expect)
    # evaluated ensemble: the chosen columns of every state on every rank, the complete rows of the own shard locally
    whole = ref.solve_evaluated(targets, geom_pos=gpos, geom_row_param=gparam, steps_per_geometry=steps, output="none")
    want = whole.eval.reshape(targets.shape[0], -1)[:, [3, 24 + 0, 24 + 23]]
    for rank in range(2):
        lo, hi = spans[rank]
        for fill in (0, 1):
            got = torch.load(os.path.join(tmp_path, f"metrics{fill}_{rank}.pt"))
            assert torch.equal(got["metrics"], want)
            assert torch.equal(got["status"], expect.info_raw[:, 32])
            assert torch.equal(got["eval_local"], whole.eval[lo * steps : hi * steps])
            assert got["sent"] == (hi - lo) * steps * (3 * 8 + 1)
            assert got["groups"] == (2, 4) and got["same_after_exchange_only"]   # one grouped call per chunk and step


def test_single_process_ensemble_needs_no_collective():
    from open_kinematics_amd.dist import solve_sharded

    table, relative = _ensemble_inputs(3, 4)
    dp = _StandInProgram()
    positions, shard = solve_sharded(dp, relative, hardpoints=table, steps_per_geometry=4)
    assert shard.geometry_range == (0, 3) and shard.exchange_bytes_per_rank == 0 and positions.shape == (12, 5, 3)
    absolute = dp.ensemble_targets(table, relative)
    again, _ = solve_sharded(dp, absolute, hardpoints=table, steps_per_geometry=4)
    assert torch.equal(again, positions)
    with pytest.raises(ValueError):
        solve_sharded(dp, relative[:3], hardpoints=table, steps_per_geometry=4)


def test_single_process_pipeline_degenerates_to_the_local_buffer():
    pipe = GatherPipeline(4, (3,), torch.float64, "cpu")
    for k in range(3):
        pipe.begin(k).fill_(float(k))
        pipe.submit(k)
    assert torch.equal(pipe.drain(), torch.full((4, 3), 2.0, dtype=torch.float64))


def test_single_process_compact_pipeline_degenerates_to_the_local_buffer():
    pipe = FreeGatherPipeline(4, 3, 2, lambda free, out: None, torch.float64, "cpu")
    assert pipe.output == "records" and pipe.bytes_sent_per_step == 0  # one rank, no collective: the solver's own records
    for k in range(3):
        pipe.begin(k).fill_(float(k))
        pipe.submit(k)
    assert torch.equal(pipe.drain(), torch.full((4, 3, 3), 2.0, dtype=torch.float64))


def test_single_process_gather_is_identity():
    x = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    assert all_gather_rows(x, 4) is x
    with pytest.raises(ValueError):
        all_gather_rows(x, 5)


def test_bench_launches_its_own_ranks_for_gpus_n():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself (torch.distributed.run as a child, before
    anything touches a GPU) and leaves with the launcher's code; --dry-launch keeps the GPUs out of it."""
    import json
    import subprocess
    import sys

    from conftest import REPO

    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [json.loads(l) for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout  # rank 0 alone prints
    assert lines[0] == {"dry_launch": True, "n_gpus": 2, "ranks_seen": 2, "master": lines[0]["master"]}
    assert lines[0]["master"].startswith("127.0.0.1:")
    # a launcher that starts another number of ranks than --gpus says is an error, not a silent mismatch
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "started 3 rank" in (bad.stderr + bad.stdout)
