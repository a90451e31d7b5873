#!/bin/bash
# A/B of generator switches on the GPU box: every variant is an environment assignment list; the kernels are compiled in
# place (hiprtc) into a scratch cache.   bash tools/ab.sh "base:" "swz:OKX_QUAD_ACC_SWIZZLE=1" ...   (configs: $AB_CONFIGS)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CONFIGS=${AB_CONFIGS:-"c2:cold:200 c5:chained:20"}
mkdir -p "$R/gpurun_out"
for variant in "$@"; do
  name=${variant%%:*}; envs=${variant#*:}
  for cfg in $CONFIGS; do
    IFS=: read -r c m k <<< "$cfg"
    out=$(env OKX_KERNEL_CACHE=/tmp/okx_ab_$name $(echo "$envs" | tr ',' ' ') python3 "$R/tools/profile_config.py" $c $m $k 2>/dev/null | tail -1)
    echo "$name $c $m $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); r=d['$m']; print('kernel_ms %.5f value %.4g evals %.3f ok %s' % (r['kernel_ms'], r['value'], r['lm_evaluations_mean'], r['all_converged']))" "$out" 2>/dev/null || echo FAILED)"
  done
done | tee -a "$R/gpurun_out/ab.log"
