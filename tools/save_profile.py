#!/usr/bin/env python3
"""Copy the artifacts of tools/profile_bench.sh from gpurun_out/ into profiles/<round>/ under a tag.
   tools/save_profile.py r01 bench_c2_pred "note about the build" """
import json, os, shutil, sys
rnd, tag, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
O, P = "gpurun_out/profile_bench", f"profiles/{rnd}"
os.makedirs(P, exist_ok=True)
s = json.load(open(f"{O}/summary.json"))
last = lambda path: json.loads(open(path).read().strip().splitlines()[-1])
bench, under = last(f"{O}/bench.json"), last(f"{O}/bench_under_rocprof.json")
shutil.copy(f"{O}/bench.json", f"{P}/{tag}.json")
shutil.copy(f"{O}/bench_under_rocprof.json", f"{P}/{tag}_under_rocprof.json")
shutil.copy(f"{O}/trace/t_kernel_stats.csv", f"{P}/{tag}_kernel_stats.csv")
with open(f"{O}/trace/t_kernel_trace.csv") as f, open(f"{P}/{tag}_kernel_trace_head.csv", "w") as g:
    g.writelines(line for i, line in enumerate(f) if i < 40)
for src, name in (("fetch/f", "FETCH_SIZE"), ("write/w", "WRITE_SIZE"), ("sq/sq", "SQ")):
    shutil.copy(f"{O}/{src}_counter_collection.csv", f"{P}/{tag}_pmc_{name}.csv")
fetch = s["FETCH_SIZE"]["per_dispatch_kib_median"] * 1024
write = s["WRITE_SIZE"]["per_dispatch_kib_median"] * 1024
sq, w = s["SQ"], s["SQ"]["SQ_WAVES"]
out = {
    "kernel": "okx_quad_solve_u (runtime-specialised quad kernel of the DW corner program)" + (": " + note if note else ""),
    "workload": "bench.py C2 16384-step sweep, one launch = 16384 problems = 1024 wavefronts (1 per SIMD)",
    "command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) --output-format csv "
               "-- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline; per-dispatch MEDIANS (the predictor's 8-problem node solve is "
               "one more, tiny dispatch of the same kernel)",
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
    "corrections": "counter unit = KiB (x1024). The gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md applies to wide (16 B/lane) "
                   "coalesced streaming reads; this kernel reads 8-byte targets and L2-resident parameter / coefficient tables, so FETCH_SIZE "
                   "is reported uncorrected (uncalibrated width per the guide; doubling it would add 0.4 MB). WRITE_SIZE is exact: 6400 KiB = "
                   "16384 x (360 B positions + 40 B info).",
    "algorithmic_bytes_per_launch": 392 * 16384,
    "sq_counters_per_wavefront": {
        "wave_cycles": sq["SQ_WAVE_CYCLES"] * 4 / w, "valu_instructions": sq["SQ_INSTS_VALU"] / w, "salu_instructions": sq["SQ_INSTS_SALU"] / w,
        "valu_active_cycles": sq["SQ_ACTIVE_INST_VALU"] * 4 / w, "wait_any_cycles": sq["SQ_WAIT_ANY"] * 4 / w,
        "wait_inst_any_cycles": sq["SQ_WAIT_INST_ANY"] * 4 / w,
        "note": "SQ cycle counters are in quad-cycles (x4 applied); 1024 wavefronts per launch"},
    "kernel_stats": s["kernel_stats"],
    "bench_value": bench["value"], "bench_kernel_ms": bench["roofline"]["kernel_ms"],
    "under_rocprof_kernel_ms": under["roofline"]["kernel_ms"],
    "raw": {k: s[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ")},
}
json.dump(out, open(f"{P}/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_bytes_per_launch", "sq_counters_per_wavefront", "kernel_stats", "bench_value",
                                      "bench_kernel_ms", "under_rocprof_kernel_ms")}, indent=1))
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "roofline", "compute", "cpu_baseline")}, indent=1))
