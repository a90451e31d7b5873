"""
Multi-GPU: the batch (sweep steps x geometries) shards by contiguous index range, one process
per GPU; there is no data-path collective while solving.  The single exchange step is an
all-gather of the solved positions (RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous, balanced ``[lo, hi)`` block of ``n_items`` for ``rank`` (first ranks get the remainder)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None, spans=None) -> torch.Tensor:
    """
    Gather per-rank row blocks into the full ``[n_total, ...]`` tensor on every rank.  The blocks are
    ``shard_range`` of the rows unless ``spans`` gives every rank's ``[lo, hi)`` explicitly (geometry-major
    shards: whole geometries per rank).  Uneven blocks are padded to the largest one for the collective.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError("single-process gather expects the full batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if spans is None:
        spans = [shard_range(n_total, r, world) for r in range(world)]
    if len(spans) != world or spans[0][0] != 0 or spans[-1][1] != n_total or \
            any(a[1] != b[0] for a, b in zip(spans, spans[1:])):
        raise ValueError("spans must tile [0, n_total) in rank order")
    lo, hi = spans[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: expected {hi - lo} rows, got {local.shape[0]}")
    biggest = max(b - a for a, b in spans)
    if hi - lo < biggest:
        pad = torch.zeros((biggest - (hi - lo), *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    gathered = torch.empty((world * biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    if n_total == world * biggest:
        return gathered
    return torch.cat([gathered[r * biggest : r * biggest + (b - a)] for r, (a, b) in enumerate(spans)], dim=0)


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

    def __init__(self, rows_per_rank: int, tail_shape, dtype, device, group=None, depth: int = 2,
                 collective_at_world_one: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # one rank with an initialised process group can still run the collective (a one-GPU box rehearsing the RCCL
        # path: communicator set-up, stream ordering of the pipeline, the expand on the gathered buffer)
        self.exchange = self.world > 1 or (collective_at_world_one and dist.is_available() and dist.is_initialized())
        self.depth = depth
        self.local = [torch.empty((rows_per_rank, *tail_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.full = [torch.empty((rows_per_rank * self.world, *tail_shape), dtype=dtype, device=device)
                     if self.exchange else None for _ in range(depth)]
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
        if self.exchange:
            self.work[slot] = dist.all_gather_into_tensor(self.full[slot], self.local[slot], group=self.group,
                                                          async_op=True)

    def drain(self) -> torch.Tensor:
        for slot in range(self.depth):
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None
        if self.last < 0:
            raise RuntimeError("nothing was submitted")
        return self.full[self.last] if self.exchange else self.local[self.last]


class FreeGatherPipeline:
    """
    The pipelined exchange of the bench step with the compact payload: the solve writes the FREE coordinates of its
    problems (``3 n_free`` doubles each, ``okx_solve_opts.output = OKX_OUTPUT_FREE``) straight into the send buffer, the
    all-gather of step ``k`` runs on the collective's stream while step ``k + 1`` solves into the other slot, and one
    ``expand`` per step rebuilds every rank's output records from the gathered block (fixed points from the design
    state, derived points re-evaluated: bit-identical to what the solver would have written) - 144 B instead of 360 B
    per double-wishbone solve on the links, no packing pass, no second copy of the local records.

        pipe = FreeGatherPipeline(rows_per_rank, n_out, n_free, expand, dtype, device)
        launches = [plan(out=buf, output=pipe.output) for buf in pipe.solve_buffers]
        for k in range(steps):
            pipe.begin(k)                          # the slot's previous exchange is complete, its buffers are free
            launches[k % len(launches)]()          # solve into pipe.solve_buffers[k % depth]
            pipe.submit(k)                         # asynchronous all-gather of what was just solved
        positions = pipe.drain()                   # [rows_per_rank * world, n_out, 3] of the last step

    ``expand(free [R, n_free, 3], out [R, n_out, 3])`` fills ``out`` (``DeviceProgram.expand``).  With one rank and no
    collective the solve writes its records itself (``output == "records"``) and ``drain`` returns them.
    """

    def __init__(self, rows_per_rank: int, n_out: int, n_free: int, expand, dtype, device, group=None, depth: int = 2,
                 collective_at_world_one: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # one rank with an initialised process group can still run the collective (a one-GPU box rehearsing the RCCL
        # path: communicator set-up, stream ordering of the pipeline, the expand on the gathered buffer)
        self.exchange = self.world > 1 or (collective_at_world_one and dist.is_available() and dist.is_initialized())
        self.depth = depth
        self.expand = expand
        self.output = "free" if self.exchange else "records"
        width = n_free if self.exchange else n_out
        self.solve_buffers = [torch.empty((rows_per_rank, width, 3), dtype=dtype, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((rows_per_rank * self.world, n_free, 3), dtype=dtype, device=device)
                         if self.exchange else None for _ in range(depth)]
        self.full = [torch.empty((rows_per_rank * self.world, n_out, 3), dtype=dtype, device=device)
                     if self.exchange else None for _ in range(depth)]
        self.work = [None] * depth
        self.pending = [False] * depth  # gathered free coordinates not expanded yet
        self.last = -1
        self.bytes_sent_per_step = rows_per_rank * n_free * 3 * torch.empty((), dtype=dtype).element_size() if self.exchange else 0

    def _finish(self, slot: int) -> None:
        if self.work[slot] is not None:
            self.work[slot].wait()  # NCCL: the current stream waits; gloo: the host does
            self.work[slot] = None
        if self.pending[slot]:
            self.expand(self.gathered[slot], self.full[slot])
            self.pending[slot] = False

    def begin(self, k: int) -> torch.Tensor:
        slot = k % self.depth
        self._finish(slot)  # the previous exchange of this slot: the positions of that step are complete now
        return self.solve_buffers[slot]

    def submit(self, k: int) -> None:
        slot = k % self.depth
        self.last = slot
        if self.exchange:
            self.work[slot] = dist.all_gather_into_tensor(self.gathered[slot], self.solve_buffers[slot], group=self.group,
                                                          async_op=True)
            self.pending[slot] = True

    def drain(self) -> torch.Tensor:
        for slot in range(self.depth):
            self._finish(slot)
        if self.last < 0:
            raise RuntimeError("nothing was submitted")
        return self.full[self.last] if self.exchange else self.solve_buffers[self.last]


@dataclass
class EnsembleShard:
    """What ``solve_sharded(..., hardpoints=...)`` returns next to the gathered positions."""

    local: object                      # BatchResult of this rank's geometries (exchange="free" with N > 1: free coordinates, no records)
    geometry_range: tuple              # [lo, hi) of the geometries this rank solved
    free_full: torch.Tensor | None     # gathered free coordinates [G * S, n_free, 3] (exchange="free")
    info_full: torch.Tensor | None     # gathered okx_info records [G * S, 40] uint8 (None with info="status")
    exchange_bytes_per_rank: int       # payload this rank contributed to the all-gather(s)
    status_full: torch.Tensor | None = None  # info="status": one byte per solve [G * S], the low byte of okx_info.flags


@dataclass
class _LocalShard:
    """This rank's part of a pipelined ensemble, shaped like the ``BatchResult`` of a compact solve."""

    free: torch.Tensor
    info_raw: torch.Tensor
    positions: object = None


def _world(group):
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    return world, (dist.get_rank(group) if world > 1 else 0)


def chunk_pieces(n_geom: int, world: int, chunks: int) -> list:
    """
    The piece table of a chunked ensemble exchange, the same on every rank: ``pieces[k][r] = (g_lo, g_hi)`` - the
    geometries rank ``r`` solves in chunk ``k`` (its ``shard_range`` block cut into ``chunks`` contiguous runs; runs may
    be empty when a rank has fewer geometries than chunks).
    """
    table = []
    for k in range(chunks):
        row = []
        for r in range(world):
            lo, hi = shard_range(n_geom, r, world)
            a, b = shard_range(hi - lo, k, chunks)
            row.append((lo + a, lo + b))
        table.append(row)
    return table


class ShardedEnsemble:
    """
    BASELINE config 5 on N GPUs as a PIPELINE (geometry-major contiguous shards, one exchange of the
    solved states).  Every rank cuts its shard into ``chunks`` runs of whole geometries and, chunk by chunk,

      * solves chunk k on the compute stream - the solve writes the free coordinates (``okx_solve_opts.output = free``, the
        exchange's payload) and the info records straight into THEIR FINAL PLACE in the gathered arrays: nothing is packed,
        staged or copied on the sending side;
      * exchanges chunk k with every peer in ONE grouped point-to-point call (``batch_isend_irecv``: on RCCL a single
        ncclGroup, every peer over its own xGMI link at once, coordinates and info records together; the receives land in
        their final place too) while chunk k + 1 solves;
      * optionally (``records=True``) rebuilds the output records of chunk k's pieces on a third stream while chunk k + 1
        travels and chunk k + 2 solves.

    A step therefore costs about ``max(solve, exchange, expand)`` of the whole batch plus one chunk of each, instead of
    their sum.  ``records=False`` returns the gathered free coordinates (what a consumer that evaluates metrics per rank,
    or writes per-rank result files, needs) and skips the expand altogether.  ``info="status"`` exchanges ONE status byte
    per solve (the low byte of ``okx_info.flags``: converged / residual exceeded / failed / ill-conditioned) instead of the
    40-byte record - 145 instead of 184 bytes per double-wishbone solve on the links; the full records of the rank's own
    shard stay in ``info_local``.  Chunked and unchunked runs give the same bits: a chunk is just a smaller launch of the
    same independent solves.

    ``metric_columns=[(metric, target), ...]`` turns the ensemble into an EVALUATED one (the program must have
    ``enable_evaluation`` done): every rank runs the evaluated solve on its shard - solve, tangents, metric catalog and
    derivative columns in one launch, no positions written (``okx_solve_evaluated_batch``) - and what travels is the
    chosen columns alone, 8 bytes each per state, beside the status byte: ``(metric, None)`` the value of a metric (name
    from ``metrics.METRIC_NAMES`` or column index of the evaluation row), ``(metric, t)`` its derivative along target
    ``t``.  ``step()`` then returns ``metric_full [G * S, K]``; the rank's own complete evaluation rows stay in
    ``eval_local [n_local, 1 + T, 24]``.  This is the form of the sharded ensemble that scales: a sensitivity study wants
    camber gain and bump steer of every perturbed geometry, not 144 bytes of coordinates per state on every rank.
    """

    def __init__(self, device_program, hardpoints, targets, steps_per_geometry: int, *, group=None, chunks: int | None = None,
                 records: bool = True, relative_targets: bool = True, info: str = "full", direct: bool | None = None,
                 metric_columns=None, **solve_kw):
        if metric_columns is not None:
            records, info = False, "status"   # nothing of the positions travels or is written
        self.dp = device_program
        self.group = group
        self.world, self.rank = _world(group)
        self.steps = int(steps_per_geometry)
        self.records = bool(records)
        program = device_program.program
        self.n_geom = int(hardpoints.shape[0])
        self.n_total = self.n_geom * self.steps
        glo, ghi = shard_range(self.n_geom, self.rank, self.world)
        self.geometry_range = (glo, ghi)
        if chunks is None:
            # Up to 8 chunks, but never a chunk below ONE FULL ROUND of the solve kernel over the chip (one 64-problem
            # wavefront on each of the 4 x CUs SIMDs: 65536 problems on an MI355X) - a smaller launch takes the same ~45 us
            # with most SIMDs idle, and the pipeline loses more on its solve stage than it hides of the exchange
            # (tools/c5_pipeline_model.py: a rank's 131072 problems at N = 8 solve in 0.09 ms as one launch or two halves, in
            # 0.40 ms as 8 eighths).  Without a GPU (the CPU tests' stand-in) a round is 4096.
            dev = torch.as_tensor(hardpoints).device
            one_round = torch.cuda.get_device_properties(dev).multi_processor_count * 256 if dev.type == "cuda" else 4096
            # (the evaluated ensemble sends 33 B instead of 145 B per state and computes 1.5 x as long: its exchange hides
            #  behind the solve with two chunks, more only lengthen the solve - tools/c5_pipeline_model.py)
            most = 2 if metric_columns is not None else 8
            # (from the SMALLEST shard, a rank-independent number: the piece tables and the grouped send / receive calls
            #  must have the same chunk count on every rank, and uneven shards differ by one geometry)
            least = min(hi - lo for lo, hi in (shard_range(self.n_geom, r, self.world) for r in range(self.world)))
            chunks = max(1, min(most, (least * self.steps) // one_round, least)) if self.world > 1 else 1
        self.chunks = max(1, int(chunks))
        # Kernel family and chain length are chosen ONCE, for the whole ensemble as one launch on one GPU (auto selection
        # goes by the problem count: a 16384-problem chunk of a million-problem ensemble would otherwise run the quad kernel
        # where the ensemble runs the lane kernel - same answers to 1e-9, other bits): every chunk on every rank is forced
        # to that choice, so chunked, unchunked, one-GPU and N-GPU runs of an ensemble agree bit for bit.
        if hasattr(device_program, "plan_launch"):
            if hasattr(device_program, "wait_ready"):
                device_program.wait_ready()  # (a program still on its interpreter kernels would pin every chunk to them)
            kernel, chain_len = device_program.plan_launch(self.n_total, steps_per_geometry=self.steps, geometry_tables=True,
                                                           evaluated=metric_columns is not None, **solve_kw)
            solve_kw = {**solve_kw, "kernel": kernel, "chain_len": chain_len}
        self.solve_kw = solve_kw
        self.pieces = chunk_pieces(self.n_geom, self.world, self.chunks)
        # per-geometry emission: the expand on the receiving side needs every geometry's fixed points (a small table,
        # replicated as SURVEY.md section 8e allows); without records only this rank's slice is rebound
        everyone = self.records and self.world > 1
        table = hardpoints if everyone else hardpoints[glo:ghi]
        gpos, gparam = device_program.rebind(table)
        self.gpos_all = gpos if everyone else None
        self.my_pos = gpos[glo:ghi] if everyone else gpos
        self.my_param = gparam[glo:ghi] if everyone else gparam
        targets = torch.as_tensor(targets)
        if relative_targets:
            self.local_targets = device_program.ensemble_targets(self.my_pos, targets)
        else:
            self.local_targets = targets[glo * self.steps : ghi * self.steps]
        device = self.device = self.my_pos.device
        if info not in ("full", "status"):
            raise ValueError("info must be 'full' or 'status'")
        self.metric_index = None
        if metric_columns is not None:
            from .metrics import METRIC_NAMES

            # (columns per evaluation row: 24 for a corner, 64 for a composed axle - left block | right block | axle metrics | roles)
            self.eval_columns = int(getattr(device_program, "eval_columns", 0) or 24)
            flat = []
            for metric, target in metric_columns:
                col = METRIC_NAMES.index(metric) if isinstance(metric, str) else int(metric)
                row = 0 if target is None else 1 + int(target)
                if not (0 <= col < self.eval_columns and 0 <= row <= program.n_targets):
                    raise ValueError(f"no evaluation entry ({metric!r}, {target!r})")
                flat.append(row * self.eval_columns + col)
            if not flat:
                raise ValueError("metric_columns is empty")
            self.metric_index = torch.tensor(flat, dtype=torch.int64, device=device)

        self.status_only = info == "status" and self.world > 1  # (one rank exchanges nothing: full records, a view of their flag byte)
        # ONE rank that wants records has nothing to exchange: its solves write the records themselves (`direct`; the gathered
        # free coordinates are then not kept - direct=False keeps the two-stage form, e.g. to compare the stages' bits)
        self.direct = (self.world == 1 and self.records and not self.status_only) if direct is None else bool(direct)
        if self.direct and (self.world > 1 or not self.records or self.status_only):
            raise ValueError("direct records need a world of one, records=True and info='full'")
        if self.metric_index is not None:
            if self.direct:
                raise ValueError("direct records and metric_columns exclude each other")
            n_local = (ghi - glo) * self.steps
            self.eval_local = torch.empty((n_local, 1 + program.n_targets, self.eval_columns), dtype=torch.float64, device=device)
            self.metric_full = torch.empty((self.n_total, len(self.metric_index)), dtype=torch.float64, device=device)
        self.free_full = None if self.metric_index is not None else torch.empty((self.n_total, program.n_free, 3), dtype=torch.float64, device=device)
        # what travels beside the coordinates: the 40-byte info records, or one status byte per solve (then the records of
        # this rank's own shard are kept in `info_local`)
        self.info_full = None if self.status_only else torch.empty((self.n_total, 40), dtype=torch.uint8, device=device)
        self.status_full = torch.empty((self.n_total,), dtype=torch.uint8, device=device) if self.status_only else None
        self.info_local = torch.empty(((ghi - glo) * self.steps, 40), dtype=torch.uint8, device=device) if self.status_only else None
        self.positions = torch.empty((self.n_total, program.n_out, 3), dtype=torch.float64, device=device) if self.records else None
        self.expand_stream = torch.cuda.Stream(device=device) if device.type == "cuda" and self.records and not self.direct else None
        if self.direct:
            self.free_full = None
        if info == "status" and not self.status_only:
            self.status_full, self.info_local = self.info_full[:, 32], self.info_full
        payload = program.n_free * 24 if self.metric_index is None else 8 * len(self.metric_index)
        self.exchange_bytes_per_rank = (ghi - glo) * self.steps * (payload + (1 if self.status_only else 40)) if self.world > 1 else 0
        # host time per chunk: the solve as a pre-bound launch (DeviceProgram.plan), the expands of a chunk's pieces (one per
        # rank) as ONE HIP graph replayed from the second step on
        self._plans = {}
        self._expand_graphs = {}
        self.use_graphs = device.type == "cuda"
        self.p2p_groups = self.p2p_ops = 0  # grouped point-to-point calls / operations issued so far

    def _rows(self, span):
        return slice(span[0] * self.steps, span[1] * self.steps)

    def _solve_chunk(self, k: int) -> None:
        glo = self.geometry_range[0]
        a, b = self.pieces[k][self.rank]
        if b <= a:
            return
        rows = self._rows((a, b))
        local = slice((a - glo) * self.steps, (b - glo) * self.steps)
        if self.metric_index is not None:
            self._solve_chunk_evaluated(k, a, b, rows, local)
            return
        out = self.positions[rows] if self.direct else self.free_full[rows]
        info = self.info_local[local] if self.status_only else self.info_full[rows]
        shape = "records" if self.direct else "free"
        if out.is_cuda and hasattr(self.dp, "plan"):
            stream = torch.cuda.current_stream(out.device).cuda_stream
            bound = self._plans.get(k)
            if bound is None or bound[0] != stream:  # (a plan launches on the stream that was current when it was made)
                bound = (stream, self.dp.plan(self.local_targets[local], geom_pos=self.my_pos[a - glo : b - glo],
                                              geom_row_param=self.my_param[a - glo : b - glo], steps_per_geometry=self.steps,
                                              output=shape, out=out, info_out=info, **self.solve_kw))
                self._plans[k] = bound
            bound[1]()
            if self.status_only:
                self.status_full[rows] = info[:, 32]
            return
        res = self.dp.solve(self.local_targets[local], geom_pos=self.my_pos[a - glo : b - glo], geom_row_param=self.my_param[a - glo : b - glo],
                            steps_per_geometry=self.steps, output=shape, out=out, info_out=info, **self.solve_kw)
        # (a stand-in program of the CPU tests returns fresh tensors instead of filling the buffers it was given)
        got = res.positions if self.direct else res.free
        if got.data_ptr() != out.data_ptr():
            out.copy_(got)
        if res.info_raw.data_ptr() != info.data_ptr():
            info.copy_(res.info_raw)
        if self.status_only:
            self.status_full[rows] = info[:, 32]  # (okx_info.flags is the int32 at byte 32: its low byte carries every flag)

    def _solve_chunk_evaluated(self, k: int, a: int, b: int, rows, local) -> None:
        """The evaluated solve of one chunk (no positions written) and its chosen columns into their place in the gathered table."""
        glo = self.geometry_range[0]
        info = self.info_local[local] if self.status_only else self.info_full[rows]
        ev = self.eval_local[local]
        kw = dict(geom_pos=self.my_pos[a - glo : b - glo], geom_row_param=self.my_param[a - glo : b - glo], steps_per_geometry=self.steps,
                  output="none", info_out=info, eval_out=ev, **self.solve_kw)
        if ev.is_cuda and hasattr(self.dp, "plan_evaluated"):
            stream = torch.cuda.current_stream(ev.device).cuda_stream
            bound = self._plans.get(k)
            if bound is None or bound[0] != stream:
                bound = (stream, self.dp.plan_evaluated(self.local_targets[local], **kw))
                self._plans[k] = bound
            bound[1]()
        else:
            res = self.dp.solve_evaluated(self.local_targets[local], **kw)
            if res.eval.data_ptr() != ev.data_ptr():  # (the CPU tests' stand-in returns fresh tensors)
                ev.copy_(res.eval)
            if res.info_raw.data_ptr() != info.data_ptr():
                info.copy_(res.info_raw)
        torch.index_select(ev.view(ev.shape[0], -1), 1, self.metric_index, out=self.metric_full[rows])
        if self.status_only:
            self.status_full[rows] = info[:, 32]

    def _exchange_chunk(self, k: int) -> list:
        if self.world == 1:
            return []
        if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
            return self._exchange_chunk_through_the_host(k)
        ops = []
        mine = self.pieces[k][self.rank]
        for peer in range(self.world):
            if peer == self.rank:
                continue
            dst = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            theirs = self.pieces[k][peer]
            beside = self.status_full if self.status_only else self.info_full
            payload = self.free_full if self.metric_index is None else self.metric_full
            if mine[1] > mine[0]:
                ops.append(dist.P2POp(dist.isend, payload[self._rows(mine)], dst, self.group))
                ops.append(dist.P2POp(dist.isend, beside[self._rows(mine)], dst, self.group))
            if theirs[1] > theirs[0]:
                ops.append(dist.P2POp(dist.irecv, payload[self._rows(theirs)], dst, self.group))
                ops.append(dist.P2POp(dist.irecv, beside[self._rows(theirs)], dst, self.group))
        if ops:  # (what a bench line / a test reports: grouped calls issued and operations inside them, since construction)
            self.p2p_groups += 1
            self.p2p_ops += len(ops)
        return dist.batch_isend_irecv(ops) if ops else []

    def _exchange_chunk_through_the_host(self, k: int) -> list:
        """Device tensors over gloo (two ranks rehearsing on one GPU: gloo's point-to-point calls take host tensors): the
        same pieces, staged through host copies, complete on return."""
        mine = self.pieces[k][self.rank]
        ops, landing = [], []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            dst = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            theirs = self.pieces[k][peer]
            for full in (self.free_full if self.metric_index is None else self.metric_full, self.status_full if self.status_only else self.info_full):
                if mine[1] > mine[0]:
                    ops.append(dist.P2POp(dist.isend, full[self._rows(mine)].cpu(), dst, self.group))
                if theirs[1] > theirs[0]:
                    host = torch.empty_like(full[self._rows(theirs)], device="cpu")
                    ops.append(dist.P2POp(dist.irecv, host, dst, self.group))
                    landing.append((full[self._rows(theirs)], host))
        if ops:
            self.p2p_groups += 1
            self.p2p_ops += len(ops)
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for device_rows, host in landing:
            device_rows.copy_(host)
        return []

    def _expand_chunk(self, k: int, works: list) -> None:
        def run():
            for w in works:
                w.wait()  # NCCL: this stream waits for the transfers; gloo: the host does
            pieces()

        def pieces():
            for r in range(self.world):
                a, b = self.pieces[k][r]
                if b <= a:
                    continue
                rows = self._rows((a, b))
                gp = self.gpos_all[a:b] if self.gpos_all is not None else self.my_pos[a - self.geometry_range[0] : b - self.geometry_range[0]]
                got = self.dp.expand(self.free_full[rows], out=self.positions[rows], geom_pos=gp, steps_per_geometry=self.steps)
                if got.data_ptr() != self.positions[rows].data_ptr():
                    self.positions[rows].copy_(got)

        if self.expand_stream is None:
            run()
            return
        self.expand_stream.wait_stream(torch.cuda.current_stream(self.expand_stream.device))  # (the own piece was solved there)
        with torch.cuda.stream(self.expand_stream):
            graph = self._expand_graphs.get(k)
            if not self.use_graphs or self.world == 1:
                run()
            elif graph is None:
                run()  # the first step: plain launches (nothing lazy inside a capture), the second step captures
                self._expand_graphs[k] = "warm"
            else:
                for w in works:
                    w.wait()
                if graph == "warm":
                    try:
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=self.expand_stream, capture_error_mode="thread_local"):
                            pieces()  # (a linear graph: forking the pieces onto side streams inside the capture made the
                            #  replay SLOWER on this runtime - 0.39 against 0.31 ms for 16 nodes, EXPERIMENTS.md section 2)
                    except Exception:  # capture refused: stay with plain launches
                        self.use_graphs = False
                        torch.cuda.synchronize(self.expand_stream.device)
                        pieces()
                        return
                    self._expand_graphs[k] = graph
                graph.replay()

    def exchange_only(self):
        """The exchange stage alone (every chunk's grouped point-to-point call on whatever the buffers hold, waited for): what
        a step costs beyond its compute when nothing overlaps - a bench figure, not part of a step."""
        pending = []
        for k in range(self.chunks):
            pending += self._exchange_chunk(k)
        for w in pending:
            w.wait()

    def step(self):
        """One pass over the whole ensemble: returns ``positions`` (``records=True``) or the gathered free coordinates."""
        pending = []
        for k in range(self.chunks):
            self._solve_chunk(k)
            works = self._exchange_chunk(k)
            if self.records and not self.direct:
                self._expand_chunk(k, works)
            else:
                pending += works
        for w in pending:
            w.wait()
        if self.expand_stream is not None:
            torch.cuda.current_stream(self.expand_stream.device).wait_stream(self.expand_stream)
        if self.metric_index is not None:
            return self.metric_full
        return self.positions if self.records else self.free_full


def solve_sharded(device_program, targets_full: torch.Tensor, gather=True, group=None, *, hardpoints=None,
                  steps_per_geometry: int = 0, exchange: str = "free", relative_targets: bool | None = None,
                  chunks: int | None = None, info: str = "full", **solve_kw):
    """
    Solve this rank's index block and (optionally) all-gather the solved positions.

    Without ``hardpoints``: ``targets_full [B, T]`` absolute targets of one geometry, sharded by contiguous index
    range; returns ``(positions, local_result)``.

    With ``hardpoints [G, P, 3]`` (BASELINE config 5: perturbed geometries x sweep steps): the
    batch is **geometry-major**, a rank owns whole geometries ``shard_range(G, rank, world)``.  The rank rebinds
    its slice of the hardpoint table on the device (per-geometry problem emission, reference
    ``suspensions/corner/double_wishbone.py:259-308``), builds its targets, solves, and the exchange step gathers
    ``exchange="free"`` the free coordinates (``3 n_free`` doubles per solve; every rank then rebuilds all positions
    with ``expand`` from the replicated, rebound hardpoint table) or ``"positions"`` the output records
    themselves, plus the 40-byte info records.  ``targets_full`` is either ``[S, T]`` RELATIVE displacements
    applied to every geometry's own design coordinates (the reference's relative target mode) or ``[G * S, T]``
    absolute values (``relative_targets`` says which; ``None`` infers it from the row count, absolute when both
    fit); ``steps_per_geometry`` = S.  Returns ``(positions [G * S, n_out, 3], EnsembleShard)``.
    The compact exchange runs as a pipeline (``ShardedEnsemble``): the shard is cut into ``chunks`` runs of whole
    geometries (default: up to 8), chunk k + 1 solves while chunk k travels - coordinates and info records in one grouped
    point-to-point call, straight from and into their final place - and chunk k - 1 is expanded on a third stream.
    ``gather="free"`` skips the expand and returns the gathered free coordinates ``[G * S, n_free, 3]`` instead of the
    records (``EnsembleShard.free_full`` holds them either way); ``info="status"`` exchanges one status byte per solve
    instead of the 40-byte record (``EnsembleShard.status_full``; ``local.info_raw`` keeps this rank's full records).
    Uneven geometry counts need no padding.
    """
    world, rank = _world(group)
    if hardpoints is None:
        n_total = targets_full.shape[0]
        lo, hi = shard_range(n_total, rank, world)
        result = device_program.solve(targets_full[lo:hi], **solve_kw)
        positions = all_gather_rows(result.positions, n_total, group) if gather else result.positions
        return positions, result
    if exchange not in ("free", "positions"):
        raise ValueError("exchange must be 'free' or 'positions'")
    if gather == "free" and exchange != "free":
        raise ValueError("gather='free' returns the compact exchange's payload: exchange must be 'free'")
    steps = int(steps_per_geometry)
    n_geom = int(hardpoints.shape[0])
    if steps <= 0:
        raise ValueError("an ensemble needs steps_per_geometry > 0")
    program = device_program.program
    glo, ghi = shard_range(n_geom, rank, world)
    targets_full = torch.as_tensor(targets_full)
    relative = relative_targets if relative_targets is not None else \
        (targets_full.shape[0] == steps and n_geom * steps != steps)
    if targets_full.shape[0] != (steps if relative else n_geom * steps):
        raise ValueError("targets must be [S, T] relative displacements or [G * S, T] absolute values")
    if gather and world > 1 and exchange == "free":
        pipe = ShardedEnsemble(device_program, hardpoints, targets_full, steps, group=group, chunks=chunks,
                               records=gather != "free", relative_targets=relative, info=info, **solve_kw)
        result = pipe.step()
        own = pipe._rows(pipe.geometry_range)
        local = _LocalShard(pipe.free_full[own], pipe.info_local if pipe.status_only else pipe.info_full[own])
        return result, EnsembleShard(local, pipe.geometry_range, pipe.free_full, pipe.info_full, pipe.exchange_bytes_per_rank,
                                     pipe.status_full)
    # Per-geometry emission.  The expand on the receiving side needs every geometry's fixed points, so with the
    # compact exchange the (small: G x P x 24 B) hardpoint table is rebound in full on every rank — replicated
    # inputs, as SURVEY.md section 8e allows; otherwise only this rank's slice.
    everyone = gather and world > 1 and exchange == "free"
    table = hardpoints if everyone else hardpoints[glo:ghi]
    gpos, gparam = device_program.rebind(table)
    my_pos = gpos[glo:ghi] if everyone else gpos
    my_param = gparam[glo:ghi] if everyone else gparam
    if relative:
        local_targets = device_program.ensemble_targets(my_pos, targets_full)
    else:
        local_targets = targets_full[glo * steps : ghi * steps]
    # compact exchange: the solve writes the free coordinates alone (okx_solve_opts.output = OKX_OUTPUT_FREE) - they ARE the
    # payload, and one expand of the gathered block rebuilds every rank's records, this rank's own included
    compact = gather and world > 1 and exchange == "free"
    result = device_program.solve(local_targets, geom_pos=my_pos, geom_row_param=my_param, steps_per_geometry=steps,
                                  **(dict(solve_kw, output="free") if compact else solve_kw))
    if not gather or world == 1:
        return result.positions, EnsembleShard(result, (glo, ghi), None, result.info_raw if gather else None, 0)
    n_total = n_geom * steps
    spans = [tuple(steps * g for g in shard_range(n_geom, r, world)) for r in range(world)]
    info_full = all_gather_rows(result.info_raw, n_total, group, spans)
    sent = result.info_raw.numel() * result.info_raw.element_size()
    if not compact:
        positions = all_gather_rows(result.positions, n_total, group, spans)
        sent += result.positions.numel() * result.positions.element_size()
        return positions, EnsembleShard(result, (glo, ghi), None, info_full, sent)
    sent += result.free.numel() * result.free.element_size()
    free_full = all_gather_rows(result.free, n_total, group, spans)
    positions = device_program.expand(free_full, geom_pos=gpos, steps_per_geometry=steps)
    return positions, EnsembleShard(result, (glo, ghi), free_full, info_full, sent)
