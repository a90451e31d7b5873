#!/bin/bash
# Instruction mix of the generated quad kernel for a BASELINE program (no GPU needed).
#   tools/quad_isa.sh dw|mac|axle [waves]
set -e
cd "$(dirname "$0")/.."
which=${1:-dw}; waves=${2:-1}
export OKX_KERNEL_CACHE=/tmp/q/cache_${which}_${waves} OKX_QUAD_WAVES=$waves
rm -rf $OKX_KERNEL_CACHE; mkdir -p $OKX_KERNEL_CACHE
python - "$which" <<'PY'
import sys
from open_kinematics_amd import _lib
from open_kinematics_amd._abi import HostProgram
from open_kinematics_amd.workloads import axle_grid_problem, bump_sweep_problem, macpherson_grid_problem
lib = _lib.load()
program, _ = {"dw": lambda: bump_sweep_problem(5), "mac": lambda: macpherson_grid_problem(4, 4), "axle": lambda: axle_grid_problem(4, 4)}[sys.argv[1]]()
rc = lib.okx_precompile(HostProgram(program).byref())
print("precompile", rc, _lib.last_error() if rc else "")
PY
tail -c +25 $OKX_KERNEL_CACHE/*.okxc > /tmp/q/k.hsaco   # strip the 24-byte cache header
/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/q/k.hsaco > /tmp/q/k.s
awk '/<okx_quad_solve_u>:/{f=1} /<okx_quad_solve_g>:/{f=0} f' /tmp/q/k.s > /tmp/q/u.s
echo "total instrs: $(grep -c '^\s*[a-z]' /tmp/q/u.s)"
for pat in v_fma_f64 v_mul_f64 v_add_f64 v_fmac_f64 v_mov_b32_dpp v_accvgpr_read v_accvgpr_write v_cndmask v_mov_b32_e32 scratch_ s_waitcnt s_nop global_load v_rcp_f64 v_rsq_f64 v_cmp v_max_f64; do echo "  $pat: $(grep -c "$pat" /tmp/q/u.s)"; done
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/q/k.hsaco | grep -E "\.name:|vgpr_count|agpr_count|sgpr_count|private_segment_fixed|vgpr_spill" | paste - - - - - - | head -5
