"""
Multi-GPU: the batch (sweep steps x geometries) shards by contiguous index range, one process
per GPU; there is no data-path collective while solving.  The single exchange step is an
all-gather of the solved positions (RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous, balanced ``[lo, hi)`` block of ``n_items`` for ``rank`` (first ranks get the remainder)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """
    Gather per-rank row blocks (sharded with ``shard_range``) into the full ``[n_total, ...]``
    tensor on every rank.  Uneven shards are padded to the largest block for the collective.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError("single-process gather expects the full batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: expected {hi - lo} rows, got {local.shape[0]}")
    biggest = -(-n_total // world)
    if hi - lo < biggest:
        pad = torch.zeros((biggest - (hi - lo), *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    gathered = torch.empty((world * biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    if n_total == world * biggest:
        return gathered
    pieces = []
    for r in range(world):
        rlo, rhi = shard_range(n_total, r, world)
        pieces.append(gathered[r * biggest : r * biggest + (rhi - rlo)])
    return torch.cat(pieces, dim=0)


class GatherPipeline:
    """
    Overlaps the exchange step with the next solve: the all-gather of step ``k`` runs on the
    collective's own stream (RCCL over xGMI) while the solve kernel of step ``k + 1`` fills the
    other of two output slots.  In steady state a step costs ``max(solve, all-gather)`` instead of
    their sum.  Equal shards only (``n_total`` divisible by the world size).

        pipe = GatherPipeline(rows_per_rank, (n_out, 3), torch.float64, device)
        for k in range(steps):
            out = pipe.begin(k)        # local buffer of this step (waits for the gather that last read it)
            ... launch the solve into ``out`` on the current stream ...
            pipe.submit(k)             # asynchronous all-gather of ``out``
        full = pipe.drain()            # gathered positions of the last step, all exchanges complete
    """

    def __init__(self, rows_per_rank: int, tail_shape, dtype, device, group=None, depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.depth = depth
        self.local = [torch.empty((rows_per_rank, *tail_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.full = [torch.empty((rows_per_rank * self.world, *tail_shape), dtype=dtype, device=device)
                     if self.world > 1 else None for _ in range(depth)]
        self.work = [None] * depth
        self.last = -1

    def begin(self, k: int) -> torch.Tensor:
        slot = k % self.depth
        if self.work[slot] is not None:
            self.work[slot].wait()  # NCCL: the current stream waits; gloo: the host does
            self.work[slot] = None
        return self.local[slot]

    def submit(self, k: int) -> None:
        slot = k % self.depth
        self.last = slot
        if self.world > 1:
            self.work[slot] = dist.all_gather_into_tensor(self.full[slot], self.local[slot], group=self.group,
                                                          async_op=True)

    def drain(self) -> torch.Tensor:
        for slot in range(self.depth):
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None
        if self.last < 0:
            raise RuntimeError("nothing was submitted")
        return self.local[self.last] if self.world == 1 else self.full[self.last]


class FreeGatherPipeline(GatherPipeline):
    """
    ``GatherPipeline`` that ships the free coordinates of each solve (``3 n_free`` doubles) instead of its output
    positions (``3 n_out``) and rebuilds the positions on the receiving side (``expand``: fixed points from the
    design state, derived points re-evaluated) — 144 B instead of 360 B per double-wishbone solve.  The exchange is
    what bounds N > 1, so the payload is what matters.

        pipe = FreeGatherPipeline(rows_per_rank, n_out, free_out_index, expand, dtype, device)
        out = pipe.begin(k); ... solve into out ...; pipe.submit(k); ...; full = pipe.drain()

    ``free_out_index``: output-list index of every free point; ``expand(free [R, n_free, 3], out [R, n_out, 3])``
    fills ``out`` (``DeviceProgram.expand``).  With one rank it degenerates to the local buffer.
    """

    def __init__(self, rows_per_rank: int, n_out: int, free_out_index: torch.Tensor, expand, dtype, device, group=None,
                 depth: int = 2):
        super().__init__(rows_per_rank, (n_out, 3), dtype, device, group, depth)
        self.free_out_index = free_out_index
        self.expand = expand
        n_free = int(free_out_index.numel())
        self.free_local = [torch.empty((rows_per_rank, n_free, 3), dtype=dtype, device=device) for _ in range(depth)]
        self.free_full = [torch.empty((rows_per_rank * self.world, n_free, 3), dtype=dtype, device=device)
                          if self.world > 1 else None for _ in range(depth)]
        self.pending = [False] * depth  # gathered free coordinates not expanded yet

    def _finish(self, slot: int) -> None:
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        if self.pending[slot]:
            self.expand(self.free_full[slot], self.full[slot])
            self.pending[slot] = False

    def begin(self, k: int) -> torch.Tensor:
        slot = k % self.depth
        self._finish(slot)  # the previous exchange of this slot: positions of that step are complete now
        return self.local[slot]

    def submit(self, k: int) -> None:
        slot = k % self.depth
        self.last = slot
        if self.world > 1:
            torch.index_select(self.local[slot], 1, self.free_out_index, out=self.free_local[slot])
            self.work[slot] = dist.all_gather_into_tensor(self.free_full[slot], self.free_local[slot], group=self.group,
                                                          async_op=True)
            self.pending[slot] = True

    def drain(self) -> torch.Tensor:
        for slot in range(self.depth):
            self._finish(slot)
        if self.last < 0:
            raise RuntimeError("nothing was submitted")
        return self.local[self.last] if self.world == 1 else self.full[self.last]


def solve_sharded(device_program, targets_full: torch.Tensor, gather: bool = True, group=None, **solve_kw):
    """
    Solve this rank's index block of ``targets_full [B, T]`` and (optionally) all-gather the
    solved positions.  Returns ``(positions, local_result)``.
    """
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    n_total = targets_full.shape[0]
    lo, hi = shard_range(n_total, rank, world)
    result = device_program.solve(targets_full[lo:hi], **solve_kw)
    positions = all_gather_rows(result.positions, n_total, group) if gather else result.positions
    return positions, result
