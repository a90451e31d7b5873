#!/bin/bash
# A/B harness: benches every build/libokx_*.so variant in one gpurun call (chain_len 1 and -1).
for so in build/libokx_*.so; do cp $so open_kinematics_amd/libokx.so
 for cl in 1 -1; do
  echo -n "$(basename $so) chain_len=$cl: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chain-len $cl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2fM/s kernel %.3f ms evals %.2f ok=%s'%(d['value']/1e6, d['roofline']['kernel_ms'], d['config']['lm_evaluations_mean'], d['config']['all_converged']))"
 done
 [ -n "$PROFILE" ] && python tools/phase_profile.py 2>&1 | grep -E "derived|rows|normal|factor|subst|LM logic"
done
