#!/usr/bin/env python3
"""
BASELINE config 5 on N GPUs, what ONE GPU can measure of it: rank 0's side of the pipeline of dist.ShardedEnsemble at
world sizes 1, 2, 4, 8 with the exchange stubbed out (no peer exists here) - the solve of the rank's shard chunk by chunk
into the gathered arrays, and the expand of EVERY rank's pieces on the third stream - against the link model for the
exchange those chunks would ride (one xGMI link per peer, ~60 GB/s sustained per direction, ~20 us per
grouped point-to-point call).  The predicted step is the pipeline's schedule (`simulate`) with those stage times; the
table in DESIGN.md section 8 is this tool's output.

  python tools/c5_pipeline_model.py [--reps 10]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import open_kinematics_amd.dist as okd  # noqa: E402
from open_kinematics_amd.batch import DeviceProgram  # noqa: E402
from open_kinematics_amd.workloads import ensemble_problem  # noqa: E402

METRIC_COLUMNS = [("camber", None), ("camber", 1), ("roadwheel_angle", 1), (21, 1)]   # what bench.py --c5-gather metrics sends
LINK_GBS = 60.0     # sustained per direction per link (153.6 GB/s per link both ways = 76.8 per direction peak)
CALL_US = 20.0      # launch + synchronisation of one grouped point-to-point call


class RankZeroOf(okd.ShardedEnsemble):
    """Rank 0 of a world of `world`, alone: its peers' rows are filled once by a full solve, the exchange is a no-op."""

    def _exchange_chunk(self, k):
        return []


def simulate(chunks: int, s: float, x: float, e: float) -> float:
    """The pipeline's schedule with measured stage times per chunk: solves back to back on the GPU, chunk k's exchange on
    the links once it is solved and chunk k - 1 has gone, chunk k's expand on the GPU once it has arrived and the GPU is
    free (the solves come first: nothing holds them back)."""
    solved = [(k + 1) * s for k in range(chunks)]
    gpu_free, link_free, end = solved[-1], 0.0, solved[-1]
    for k in range(chunks):
        arrived = max(solved[k], link_free) + x
        link_free = arrived
        end = arrived
        if e > 0.0:
            gpu_free = max(arrived, gpu_free) + e
            end = gpu_free
    return end


def ms(fn, device, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 10
    device = torch.device("cuda:0")
    program, table, rel = ensemble_problem()
    dp = DeviceProgram(program, device)
    table = torch.as_tensor(table, device=device)
    spg = rel.shape[0]
    n_total = table.shape[0] * spg
    from open_kinematics_amd.input import load_geometry
    from open_kinematics_amd.metrics import corner_roles
    from open_kinematics_amd.workloads import geometry_path

    dp.enable_evaluation(corner_roles(load_geometry(geometry_path("geometry.yaml")), program))
    whole = okd.ShardedEnsemble(dp, table, rel, spg, chunks=1, records=False, chain_len=1, predictor=False)
    coords = whole.step().clone()
    records_ref = dp.expand(coords, geom_pos=dp.rebind(table)[0], steps_per_geometry=spg)
    rows = []
    real_world = okd._world
    for world in (1, 2, 4, 8):
        okd._world = lambda group, w=world: (w, 0)
        try:
            for mode, info in ((True, "full"), (True, "status"), (False, "status"), ("metrics", "status")):
                for chunks in ((1,) if world == 1 else (1, 2, 4, 8, None)):
                    metric_mode = mode == "metrics"
                    mkw = dict(metric_columns=METRIC_COLUMNS) if metric_mode else {}
                    records = False if metric_mode else mode
                    pipe = RankZeroOf(dp, table, rel, spg, chunks=chunks, records=records, info=info, chain_len=1, predictor=False, **mkw)
                    auto, chunks = chunks is None, pipe.chunks
                    if pipe.free_full is not None:
                        pipe.free_full.copy_(coords)   # what the peers would have sent
                    compute = ms(pipe.step, device, reps)
                    if records:  # plans and graph replays: still the same bits
                        assert torch.equal(pipe.step(), records_ref)
                    solve_only = RankZeroOf(dp, table, rel, spg, chunks=chunks, records=False, info=info, chain_len=1, predictor=False, **mkw)
                    solve = ms(solve_only.step, device, reps)
                    sent = pipe.exchange_bytes_per_rank if world > 1 else 0
                    # every peer over its own link at once: one copy of this rank's shard per link
                    exchange = (sent / (LINK_GBS * 1e9) * 1e3 + chunks * CALL_US * 1e-3) if world > 1 else 0.0
                    step = simulate(chunks, solve / chunks, exchange / chunks, max(compute - solve, 0.0) / chunks if records else 0.0)
                    rows.append({"world": world, "records": "metrics" if metric_mode else records, "info": info, "chunks": chunks, "auto": auto,
                                 "solve_ms": round(solve, 4), "compute_side_ms": round(compute, 4),
                                 "bytes_sent_per_link": sent, "exchange_model_ms": round(exchange, 4),
                                 "predicted_step_ms": round(step, 4), "predicted_solves_per_s": n_total / step * 1e3,
                                 "unpipelined_sum_ms": round(solve + exchange + (compute - solve if records else 0.0), 4)})
                    print(json.dumps(rows[-1]), flush=True)
        finally:
            okd._world = real_world
    print(json.dumps({"n_total": n_total, "link_gbs": LINK_GBS, "call_us": CALL_US, "rows": rows}))


if __name__ == "__main__":
    main()
