#!/bin/bash
# Registers, scratch, LDS and the instruction mix of the generated quad kernels of a BASELINE program (no GPU needed):
# compiles into a scratch kernel cache (developer switches in OKX_DEV are honoured, the source is kept beside the code).
#   tools/quad_isa.sh dw|mac|axle [cache dir]
set -e
cd "$(dirname "$0")/.."
which=${1:-dw}; d=${2:-/tmp/okx_isa_$which}
rm -rf "$d"; mkdir -p "$d"
OKX_KERNEL_CACHE=$d OKX_DEV="${OKX_DEV:+$OKX_DEV,}keep_source,no_lane" python3 tools/precompile.py "$which"
cd "$d"
for f in okxq*.okxc; do
  b=$(basename "$f" .okxc); tail -c +25 "$f" > "$b.hsaco"   # strip the 24-byte cache header
  /opt/rocm/lib/llvm/bin/llvm-objdump -d "$b.hsaco" > "$b.s"
  grep -q "okx_quad_solve_u" "$b.s" || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$b.hsaco" | grep -E "\.name:|\.vgpr_count|agpr_count|\.sgpr_count|private_segment_fixed|group_segment_fixed" | paste - - - - - - | grep -E "solve|cold"
  for k in okx_quad_solve_u okx_quad_cold_u; do
    awk -v k="<$k>:" '$0 ~ /^[0-9a-f]+ <.*>:/{f = index($0,k)>0} f' "$b.s" > "$k.s"
    [ -s "$k.s" ] || continue
    echo "== $k: $(grep -c '^\s*[a-z]' $k.s) instructions; fp64 $(grep -c -E 'v_(fma|mul|add|fmac)_f64' $k.s) dpp $(grep -c dpp $k.s) accvgpr $(grep -c accvgpr $k.s) cndmask $(grep -c cndmask $k.s) s_waitcnt $(grep -c s_waitcnt $k.s) global loads $(grep -c -E 'global_load|flat_load' $k.s) lane spills $(grep -c -E 'v_readlane|v_writelane' $k.s) scratch $(grep -c scratch_ $k.s)   ($d/$k.s)"
  done
done
