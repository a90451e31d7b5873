for single in 1 0; do for cap in 1 2 4 6 8; do
  echo -n "single=$single cap=$cap: "; OKX_FORCE_SINGLE=$single OKX_BLOCKS_PER_CU=$cap python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2fM/s kernel %.3f ms'%(d['value']/1e6, d['roofline']['kernel_ms']))"
done; done
