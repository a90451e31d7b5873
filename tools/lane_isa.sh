#!/bin/bash
# Instruction mix, registers and scratch of the generated LANE kernel for a BASELINE program (no GPU needed).
#   tools/lane_isa.sh dw|mac [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
which=${1:-dw}; shift || true
mkdir -p /tmp/q
python - "$which" <<'PY'
import ctypes as C, sys
from open_kinematics_amd import _lib
from open_kinematics_amd._abi import HostProgram
from open_kinematics_amd.workloads import bump_sweep_problem, macpherson_grid_problem
lib = _lib.load()
program, _ = {"dw": lambda: bump_sweep_problem(5), "mac": lambda: macpherson_grid_problem(4, 4)}[sys.argv[1]]()
hp = HostProgram(program)
lib.okx_lane_source.restype = C.c_int64
need = lib.okx_lane_source(hp.byref(), None, 0)
if need < 0:
    raise SystemExit(_lib.last_error())
buf = C.create_string_buffer(need)
lib.okx_lane_source(hp.byref(), buf, need)
open("/tmp/q/lane.hip", "wb").write(buf.value)
print("source bytes", need)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -include hip/hip_runtime.h -S -o /tmp/q/lane.s /tmp/q/lane.hip "$@" 2>&1 | grep -v warning | head -20
for k in okx_lane_solve_u okx_lane_solve_u_c okx_lane_chain_u okx_lane_eval; do
  awk -v k="$k" '$0 ~ "^"k":"{f=1} f&&/s_endpgm/{print; f=0} f' /tmp/q/lane.s > /tmp/q/lane_$k.s
  echo "== $k: $(grep -c '^\s*[a-z]' /tmp/q/lane_$k.s) instructions"
  for pat in v_fma_f64 v_mul_f64 v_add_f64 v_fmac_f64 v_accvgpr_read v_accvgpr_write v_cndmask v_mov_b32 v_readlane v_writelane scratch_ ds_read ds_write s_waitcnt global_load v_rcp_f64 v_rsq_f64 v_max_f64 v_readfirstlane; do echo "   $pat: $(grep -c "$pat" /tmp/q/lane_$k.s)"; done | paste - - - - - -
done
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):|^\s+\.name:\s+okx" /tmp/q/lane.s | paste - - - - - - - -
