#!/usr/bin/env python3
"""Copy the artifacts of tools/profile_run.sh from gpurun_out/profile_<tag>/ into profiles/<round>/ under that tag and write
   the per-launch traffic / SQ summary next to them.
   tools/save_profile.py r02 bench_c2_cold <units per launch> <algorithmic bytes per unit> "note about the build" """
import json, os, shutil, sys, glob
rnd, tag, units, bytes_per = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
note = sys.argv[5] if len(sys.argv) > 5 else ""
# FETCH_SIZE correction for this kernel's READ SHAPE, calibrated on 1.07 GB of 360-byte records (tools/micro/fetch_calib.hip,
# profiles/r06/fetch_calib.json): contiguous reads report exactly 1/2 of the bytes at 16 B AND at 8 B per lane (factor 2.000);
# one lane per record with 8-byte loads at a 360-byte lane stride reports 1/1.73 of the record bytes (factor 1.73)
fetch_factor = float(sys.argv[6]) if len(sys.argv) > 6 else 2.0
O, P = f"gpurun_out/profile_{tag}", f"profiles/{rnd}"
os.makedirs(P, exist_ok=True)
s = json.load(open(f"{O}/summary.json"))
last = lambda path: json.loads(open(path).read().strip().splitlines()[-1])
run, under = last(f"{O}/run.json"), last(f"{O}/run_under_rocprof.json")
find = lambda d, suffix: (glob.glob(f"{O}/{d}/**/*{suffix}", recursive=True) or [None])[0]
shutil.copy(f"{O}/run.json", f"{P}/{tag}.json")
shutil.copy(f"{O}/run_under_rocprof.json", f"{P}/{tag}_under_rocprof.json")
shutil.copy(find("trace", "kernel_stats.csv"), f"{P}/{tag}_kernel_stats.csv")
with open(find("trace", "kernel_trace.csv")) as f, open(f"{P}/{tag}_kernel_trace_head.csv", "w") as g:
    g.writelines(line for i, line in enumerate(f) if i < 40)
for d, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ"), ("sq2", "SQ2"), ("sq3", "SQ3")):
    src = find(d, "counter_collection.csv")
    if src:  # the first 6000 counter rows (a default bench run has > 4000 dispatches: megabytes of identical rows); the
        # per-dispatch medians in <tag>_pmc_traffic.json are taken over ALL dispatches, from summary.json
        with open(src) as f, open(f"{P}/{tag}_pmc_{name}.csv", "w") as g:
            g.writelines(line for i, line in enumerate(f) if i <= 6000)
fetch_raw = s["FETCH_SIZE"]["per_dispatch_kib_median"] * 1024
fetch = fetch_raw * fetch_factor
write = s["WRITE_SIZE"]["per_dispatch_kib_median"] * 1024
sq, w = s["SQ"], s["SQ"]["SQ_WAVES"]
kms = lambda r: r.get("roofline", {}).get("kernel_ms") or next((r[m]["kernel_ms"] for m in ("cold", "chained") if m in r), None)
out = {
    "kernel": s.get("dispatch", {}).get("Kernel_Name", "") + (": " + note if note else ""),
    "dispatch": s.get("dispatch"),
    "workload": run.get("config", {}).get("workload") or run.get("workload"),
    "units_per_launch": units,
    "command": "tools/profile_run.sh: rocprofv3 --kernel-trace --stats; then --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in "
               "SEPARATE passes, --output-format csv; per-dispatch MEDIANS over the dispatches of the solve kernel",
    "fetch_bytes_per_launch": fetch, "fetch_size_counter_bytes": fetch_raw, "fetch_correction_factor": fetch_factor,
    "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
    "corrections": "counter unit = KiB (x1024).  FETCH_SIZE x fetch_correction_factor: calibrated for this repository's read shapes "
                   "on a 1.07 GB buffer of 360-byte records (tools/micro/fetch_calib.hip -> profiles/r06/fetch_calib.json): contiguous "
                   "reads, 16 B or 8 B per lane, report exactly half of the bytes (x 2.000, the guide's gfx950 correction holds at 8 B per "
                   "lane too); one lane per record, 8-byte loads at a 360-byte lane stride: x 1.73.  WRITE_SIZE is exact for the "
                   "16-byte-per-lane record stores.",
    "algorithmic_bytes_per_launch": bytes_per * units,
    "traffic_over_algorithmic": (fetch + write) / (bytes_per * units),
    "sq_counters_per_wavefront": {
        "waves": w,
        "wave_cycles": sq["SQ_WAVE_CYCLES"] * 4 / w, "valu_instructions": sq["SQ_INSTS_VALU"] / w, "salu_instructions": sq["SQ_INSTS_SALU"] / w,
        "valu_active_cycles": sq["SQ_ACTIVE_INST_VALU"] * 4 / w, "wait_any_cycles": sq["SQ_WAIT_ANY"] * 4 / w,
        "wait_inst_any_cycles": sq["SQ_WAIT_INST_ANY"] * 4 / w, "active_inst_any_cycles": sq["SQ_ACTIVE_INST_ANY"] * 4 / w,
        "wait_any_share": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
        "note": "SQ cycle counters are in quad-cycles (x4 applied)"},
    "sq2_per_wavefront": {k: v / w for k, v in s.get("SQ2", {}).items()},
    "fp64_instructions_per_launch": (lambda q: None if not q or "SQ_INSTS_VALU_FMA_F64" not in q else {
        "add": q.get("SQ_INSTS_VALU_ADD_F64", 0.0), "mul": q.get("SQ_INSTS_VALU_MUL_F64", 0.0), "fma": q["SQ_INSTS_VALU_FMA_F64"],
        "trans": q.get("SQ_INSTS_VALU_TRANS_F64", 0.0), "valu_total": q.get("SQ_INSTS_VALU"),
        "flops_64_lanes": 64.0 * (q.get("SQ_INSTS_VALU_ADD_F64", 0.0) + q.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2.0 * q["SQ_INSTS_VALU_FMA_F64"]),
        "note": "wave-level instruction counts (per-dispatch medians) x 64 lanes; v_max / v_min / compares and the 32-bit moves are not "
                "in them; one lane in four carries zeros in the quad layout, so the useful share is 3/4 of flops_64_lanes"})(s.get("SQ3")),
    "kernel_stats": s["kernel_stats"],
    "kernel_ms_events": kms(run), "kernel_ms_events_under_rocprof": kms(under),
    "raw": {k: s[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2", "SQ3") if k in s},
}
json.dump(out, open(f"{P}/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("dispatch", "traffic_bytes_per_launch", "traffic_over_algorithmic", "sq_counters_per_wavefront",
                                      "sq2_per_wavefront", "kernel_stats", "kernel_ms_events", "kernel_ms_events_under_rocprof")}, indent=1))
