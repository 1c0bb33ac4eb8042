# A/B of the weight slab allocator (VITS_NO_SLAB=1: one hipMalloc per tensor, as before)
for v in 1 0; do
  echo "== NO_SLAB=$v"
  if [ $v = 1 ]; then export VITS_NO_SLAB=1; else unset VITS_NO_SLAB; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('c2 ms', j['ms_per_step'], 'xRT', j.get('x_realtime'), 'ks_ms', r.get('kernel_ms_per_forward'), 'c3 ms', j['batch32']['ms_per_step'], 'm2 ms', j['multistream']['ms_per_step'])
print({k: v for k, v in r['by_op_ms_per_forward'].items()})
"
done
