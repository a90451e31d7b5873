"""
The host cores this process can really use, and torch's intra-op thread pool fitted to them.

A GPU box hands a one-GPU job a SHARE of the host through a cgroup CPU quota (16 of 256 hardware threads on the MI355X
pool) while `nproc` and the affinity mask still say 256.  torch then sizes its OpenMP pool to 128 threads; the first
parallel host operation (a pinned-memory copy, `.cpu()` of a result, a comparison) wakes all of them, their spin-waiting
burns the cgroup's 1.6 CPU-seconds per 100 ms period, and the kernel THROTTLES the whole process until the period ends:
one stall of 20 - 85 ms, quantised by the scheduler tick, somewhere in the next few hundred sweeps.  Measured on two
boxes (profiles/r05/EXPERIMENTS.md section 8: cpu.stat nr_throttled = 1 and throttled_usec = 67 - 85 ms in exactly the
runs that stalled, 0 in the others; with the pool at the quota, five fresh processes within 2 %).  The solve kernels do
not care, a host-synchronised sweep loop of ~170 us per sweep does.
"""

from __future__ import annotations

import os

_fitted = None


def host_cores() -> tuple:
    """(cores, how): the affinity mask, cut to the cgroup's CPU quota when there is one.  No other cap."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    how = f"affinity mask of {cores}"
    try:
        with open("/sys/fs/cgroup/cpu.max", "r", encoding="utf-8") as fh:   # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            share = max(1, int(round(int(quota) / int(period))))
            if share < cores:
                cores, how = share, f"cgroup CPU quota of {share} inside an affinity mask of {cores}"
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r", encoding="utf-8") as fq, \
                    open("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r", encoding="utf-8") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0 and max(1, round(quota / period)) < cores:
                share = max(1, int(round(quota / period)))
                cores, how = share, f"cgroup CPU quota of {share} inside an affinity mask of {cores}"
        except (OSError, ValueError):
            pass
    return cores, how


def fit_host_threads(reserve: int = 2, processes: int | None = None) -> dict:
    """Cut torch's intra-op pool to the cores the cgroup grants - shared between `processes` ranks of one node (default:
    LOCAL_WORLD_SIZE, else 1), minus `reserve` for the launching thread and the runtime's own - never raising it.
    The first call decides; `OKX_KEEP_HOST_THREADS=1` leaves the pool alone.  Returns what it did."""
    global _fitted
    if _fitted is not None:
        return _fitted
    import torch

    cores, how = host_cores()
    before = torch.get_num_threads()
    if processes is None:
        try:
            processes = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            processes = 1
    want = max(1, cores // processes - reserve)
    quota_binds = how.startswith("cgroup") or processes > 1  # (an affinity mask alone already sized torch's pool)
    if os.environ.get("OKX_KEEP_HOST_THREADS") == "1" or want >= before or not quota_binds:
        _fitted = {"torch_threads": before, "was": before, "host_cores": cores, "how": how, "changed": False}
        return _fitted
    torch.set_num_threads(want)
    _fitted = {"torch_threads": torch.get_num_threads(), "was": before, "host_cores": cores, "how": how, "changed": True}
    return _fitted
