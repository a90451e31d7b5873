#!/bin/bash
# Build and run tools/micro/fetch_calib.hip on the GPU box: plain, then under rocprofv3 with the FETCH_SIZE counter alone;
# prints the per-kernel ratio bytes actually read / FETCH_SIZE.   bash tools/micro/fetch_calib.sh
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/fetch_calib
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib "$R/tools/micro/fetch_calib.hip"
/tmp/fetch_calib > "$O/run.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- /tmp/fetch_calib > "$O/run_under_rocprof.json" 2> "$O/rocprof.err"
python3 - "$O" <<'PY'
import csv, glob, json, statistics as st, sys
O = sys.argv[1]
run = json.load(open(f"{O}/run.json"))
path = glob.glob(f"{O}/fetch/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
out = {"bytes_read_per_launch": run["bytes_read_per_launch"], "kernels": {}}
for name, timing in run["kernels"].items():
    v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith(name) and r["Counter_Name"] == "FETCH_SIZE"]
    kib = st.median(v)
    out["kernels"][name] = {**timing, "fetch_size_kib_median": kib, "dispatches": len(v),
                            "bytes_over_fetch_size": run["bytes_read_per_launch"] / (kib * 1024.0)}
json.dump(out, open(f"{O}/fetch_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
