#!/usr/bin/env python3
"""
Why the host-to-host legs of bench.py (e2e, compact, zero_copy) spread from run to run: each leg in FRESH processes, with
the placement facts next to the rate - NUMA node of the GPU, NUMA node(s) the pinned buffers' pages landed on, the CPUs the
process may run on, the PCIe link - and the same legs with the process confined to each NUMA node in turn while it
allocates (first touch decides where pinned pages live).

  python tools/e2e_spread.py [--runs 5] [--no-confine] [--env KEY=VALUE[,KEY=VALUE]]...
        parent: spawns the children (fresh processes), prints one JSON line per child + a summary; every --env adds one
        more set of runs under that environment (OKX_SPREAD_WIDE_POOL=1: torch's thread pool left as wide as nproc says,
        the state of every run before open_kinematics_amd/hostcpu.py)

What it found (profiles/r05/EXPERIMENTS.md section 8): not placement - the pinned pages always sit on the GPU's NUMA node,
the link is Gen5 x16 - but the cgroup's CPU quota: cpu.stat counts one throttle of 67 - 85 ms in exactly the runs that
stall, and none once torch's pool fits the quota.
"""
import glob
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def cpus_of_node(node: int) -> list:
    path = f"/sys/devices/system/node/node{node}/cpulist"
    if not os.path.exists(path):
        return []
    cpus = []
    for part in open(path).read().strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus += list(range(int(a), int(b) + 1))
        elif part:
            cpus.append(int(part))
    return cpus


def numa_nodes() -> list:
    return sorted(int(re.search(r"node(\d+)$", p).group(1)) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))


def pages_by_node(tensor) -> dict:
    """NUMA nodes of a host tensor's pages, from /proc/self/numa_maps (the mapping that contains its first byte)."""
    addr = tensor.data_ptr()
    best = None
    try:
        for line in open("/proc/self/numa_maps"):
            start = int(line.split()[0], 16)
            if start <= addr and (best is None or start > best[0]):
                best = (start, line)
    except OSError:
        return {}
    if best is None:
        return {}
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"\bN(\d+)=(\d+)", best[1])}


def gpu_facts() -> dict:
    import torch

    facts = {}
    props = torch.cuda.get_device_properties(0)
    bdf = None
    for fmt in ("{:04x}:{:02x}:{:02x}.0",):
        try:
            bdf = fmt.format(props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        except AttributeError:
            bdf = None
    if bdf and os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
        base = f"/sys/bus/pci/devices/{bdf}"
        for key in ("numa_node", "current_link_speed", "current_link_width", "max_link_speed", "max_link_width"):
            try:
                facts[key] = open(f"{base}/{key}").read().strip()
            except OSError:
                pass
    facts["pci"] = bdf
    return facts


def cpu_throttle() -> dict:
    """This process's cgroup CPU-bandwidth counters (cgroup v2 cpu.stat, v1 cpu,cpuacct/cpu.stat) and its quota."""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time"):
                    out[k] = int(v)
            break
        except OSError:
            continue
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            out["quota"] = open(path).read().strip()
            break
        except OSError:
            continue
    return out


def threads_now() -> int:
    try:
        return int(re.search(r"Threads:\s+(\d+)", open("/proc/self/status").read()).group(1))
    except (OSError, AttributeError):
        return -1


def child(confine: int) -> None:
    if confine >= 0:
        cpus = cpus_of_node(confine)
        if cpus:
            os.sched_setaffinity(0, set(cpus) & os.sched_getaffinity(0) or os.sched_getaffinity(0))
    if os.environ.get("OKX_SPREAD_WIDE_POOL") == "1":
        os.environ["OKX_KEEP_HOST_THREADS"] = "1"  # what every run before the fix did: torch's pool as wide as nproc says
    import torch

    import bench
    from open_kinematics_amd.batch import DeviceProgram
    from open_kinematics_amd.workloads import bump_sweep_problem

    device = torch.device("cuda:0")
    torch.cuda.set_device(0)
    program, targets_host = bump_sweep_problem(16384)
    dp = DeviceProgram(program, device)
    kw = dict(chain_len=-1, predictor=False)
    out = {"env": json.loads(os.environ.get("OKX_SPREAD_ENV", "{}")), "confined_to_node": confine, "cpus_allowed": len(os.sched_getaffinity(0)), "gpu": gpu_facts()}
    buffers = bench.zero_copy_buffers(program, targets_host)
    out["pinned_pages_by_node"] = pages_by_node(buffers[0][1])
    out["zero_copy"] = bench.measure_e2e_zero_copy(dp, targets_host, device, 200, kw, buffers=buffers)["value"]
    out["compact"] = bench.measure_e2e_compact(dp, targets_host, device, 200, kw)["value"]
    before = cpu_throttle()
    e2e = bench.measure_e2e(dp, targets_host, device, 200, kw)
    after = cpu_throttle()
    out["threads"] = threads_now()
    out["torch_threads"] = torch.get_num_threads()
    out["cgroup_cpu"] = {"quota": after.get("quota"), **{k: after[k] - before.get(k, 0) for k in after if k != "quota"}}
    out["e2e"] = e2e["value"]
    out["e2e_detail"] = {k: e2e[k] for k in ("value_at_median", "ms_per_sweep", "ms_per_sweep_median", "ms_per_sweep_max",
                                             "sweeps_over_3x_median", "slowest_sweep_index", "h2d_ms", "kernel_ms", "d2h_ms")}
    print(json.dumps(out), flush=True)


def main() -> None:
    if "--child" in sys.argv:
        child(int(sys.argv[sys.argv.index("--child") + 1]))
        return
    runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 5
    nodes = numa_nodes()
    rows = []
    # --env KEY=VALUE[,KEY=VALUE]: the same children under a runtime setting (each --env is one more set of runs)
    settings = [{}] + [dict(kv.split("=", 1) for kv in sys.argv[k + 1].split(",")) for k, a in enumerate(sys.argv) if a == "--env"]
    plan = [(-1, e) for e in settings for _ in range(runs)]
    if "--no-confine" not in sys.argv:
        plan += [(node, {}) for node in nodes for _ in range(max(2, runs // 2))]
    for confine, extra in plan:
        if True:
            proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(confine)], capture_output=True, text=True,
                                  timeout=600, env={**os.environ, **extra, "OKX_SPREAD_ENV": json.dumps(extra)})
            line = [l for l in proc.stdout.splitlines() if l.startswith("{")]
            if proc.returncode != 0 or not line:
                print(json.dumps({"confined_to_node": confine, "error": (proc.stderr or proc.stdout)[-400:]}), flush=True)
                continue
            rows.append(json.loads(line[-1]))
            print(line[-1], flush=True)
    summary = {}
    for key in sorted({(r["confined_to_node"], json.dumps(r["env"])) for r in rows}):
        mine = [r for r in rows if (r["confined_to_node"], json.dumps(r["env"])) == key]
        name = f"node {key[0]} {key[1]}"
        summary[name] = {leg: [round(r[leg] / 1e6, 1) for r in mine] for leg in ("zero_copy", "compact", "e2e")}
        summary[name]["e2e_at_median"] = [round(r["e2e_detail"]["value_at_median"] / 1e6, 1) for r in mine]
        summary[name]["e2e_slowest_sweep_ms"] = [round(r["e2e_detail"]["ms_per_sweep_max"], 1) for r in mine]
        summary[name]["e2e_slowest_sweep_index"] = [r["e2e_detail"]["slowest_sweep_index"] for r in mine]
        summary[name]["cgroup_cpu_during_e2e"] = [r.get("cgroup_cpu") for r in mine]
        summary[name]["threads"] = [(r.get("threads"), r.get("torch_threads")) for r in mine]
        summary[name]["pinned_pages_by_node"] = [r["pinned_pages_by_node"] for r in mine]
    print(json.dumps({"numa_nodes": nodes, "gpu": rows[0]["gpu"] if rows else None, "M_solves_per_s_by_confinement": summary}))


if __name__ == "__main__":
    main()
