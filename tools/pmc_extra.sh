#!/bin/bash
# One more PMC pass (instruction fetch, LDS / VMEM latency levels, branches, bank conflicts) for a command; per-dispatch
# medians of the solve kernel.   bash tools/pmc_extra.sh <tag> <python script> [args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift; SCRIPT=$R/$1; shift
O=$R/gpurun_out/pmc_extra_$TAG; rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d "$O/a" -o a -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/a.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES --output-format csv -d "$O/b" -o b -- python3 "$SCRIPT" "$@" > /dev/null 2> "$O/b.err"
python3 - "$O" <<'PY'
import csv, glob, statistics as st, sys, collections
O = sys.argv[1]
for d in ("a", "b"):
    for path in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"].startswith("okx_quad_solve"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print({k: st.median(v) for k, v in acc.items()})
PY
