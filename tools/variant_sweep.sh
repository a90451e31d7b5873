for w in 2 3 4; do cp build/libokx_w$w.so open_kinematics_amd/libokx.so
 for single in 1 0; do
  echo -n "waves=$w single=$single: "; OKX_FORCE_SINGLE=$single python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2fM/s kernel %.3f ms ok=%s'%(d['value']/1e6, d['roofline']['kernel_ms'], d['config']['all_converged']))"
 done; done
