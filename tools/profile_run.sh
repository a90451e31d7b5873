#!/bin/bash
# Run on the GPU box (through gpurun): one command plain, then under rocprofv3 --kernel-trace --stats, then the PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, with --kernel-trace only).  Everything lands under
# gpurun_out/profile_<tag>/ with a summary.json; tools/save_profile.py copies it into profiles/<round>/.
#   bash tools/profile_run.sh <tag> <kernel-name-prefix> <python script> [args...]
#   e.g. bash tools/profile_run.sh bench_c2_cold okx_quad_solve bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 10
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; KPREFIX=$2; shift 2
SCRIPT=$R/$1; shift
O=$R/gpurun_out/profile_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
python3 "$SCRIPT" "$@" > "$O/run.json" 2> "$O/run.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python3 "$SCRIPT" "$@" > "$O/run_under_rocprof.json" 2> "$O/rocprof.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq" -o sq -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/pmc_sq.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$O/sq3" -o sq3 -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/pmc_sq3.err" || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$O/sq2" -o sq2 -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/pmc_sq2.err" || true
python3 - "$O" "$KPREFIX" <<'PY'
import csv, glob, json, statistics as st, sys
O, prefix = sys.argv[1], sys.argv[2]
def find(d, suffix):
    hits = glob.glob(f"{O}/{d}/**/*{suffix}", recursive=True)
    return hits[0] if hits else None
def vals(path, name):
    if not path: return []
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Kernel_Name"].startswith(prefix) and r["Counter_Name"] == name]
out = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    v = vals(find(d, "counter_collection.csv"), name)
    # median over dispatches: setup launches of the same kernel (predictor node solves, warm-up of another mode) are outliers
    out[name] = {"per_dispatch_kib_median": st.median(v), "per_dispatch_kib_mean": st.mean(v), "min": min(v), "max": max(v), "dispatches": len(v)} if v else None
for grp, names in (("SQ", ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")),
                   ("SQ2", ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_FLAT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS")),
                   ("SQ3", ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))):
    path = find({"SQ": "sq", "SQ2": "sq2", "SQ3": "sq3"}[grp], "counter_collection.csv")
    rec = {}
    for name in names:
        v = vals(path, name)
        if v: rec[name] = st.median(v)
    out[grp] = rec
stats_path = find("trace", "kernel_stats.csv")
out["kernel_stats"] = [r for r in csv.DictReader(open(stats_path)) if r["Name"].startswith(prefix)] if stats_path else []
trace_path = find("trace", "kernel_trace.csv")
if trace_path:
    rows = [r for r in csv.DictReader(open(trace_path)) if r["Kernel_Name"].startswith(prefix)]
    if rows:
        r = rows[-1]
        out["dispatch"] = {k: r[k] for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -c 1500 "$O/run.json"
