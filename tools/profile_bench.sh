#!/bin/bash
# Run on the GPU box (through gpurun): bench.py, its rocprofv3 kernel trace and the two PMC passes for HBM
# traffic; leaves everything under gpurun_out/profile_bench/ and a summary JSON next to it.
#   gpurun -- 'bash tools/profile_bench.sh'
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profile_bench
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python3 "$R/bench.py" --steps 200 --warmup 10 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -o sq -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/pmc_sq.err"
python3 - "$O" <<'PY'
import csv, json, statistics as st, sys
O = sys.argv[1]
def vals(path, name):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Kernel_Name"].startswith("okx_quad_solve") and r["Counter_Name"] == name]
out = {}
for name, path in (("FETCH_SIZE", f"{O}/fetch/f_counter_collection.csv"), ("WRITE_SIZE", f"{O}/write/w_counter_collection.csv")):
    v = vals(path, name)
    # median: the predictor's 8-problem node solve is one more (tiny) dispatch of the same kernel
    out[name] = {"per_dispatch_kib_median": st.median(v), "per_dispatch_kib_mean": st.mean(v), "min": min(v), "max": max(v), "dispatches": len(v)}
sq = {}
for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    v = vals(f"{O}/sq/sq_counter_collection.csv", name)
    sq[name] = st.median(v)
out["SQ"] = sq
stats = [r for r in csv.DictReader(open(f"{O}/trace/t_kernel_stats.csv")) if r["Name"].startswith("okx_quad_solve")]
out["kernel_stats"] = stats
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat "$O/bench.json"
