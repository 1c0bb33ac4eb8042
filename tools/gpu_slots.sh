#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/slots; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "bf16x3 or heavy_tailed" 2>&1 | tail -3
for m in 3 2 3 2; do
  BENCH_SKIP_FINITE_CHECK=1 VITS_BF3_SLOTS=$m timeout 300 python bench.py --workload c3 --precision bf16x3 --no-cpu-baseline --no-host-api --steps 10 --warmup 3 > $O/s$m.json 2> $O/s$m.err || echo "s$m failed: $(tail -2 $O/s$m.err)"
  python - $m <<'P'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/slots/s{v}.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
    print('slots',v, d['ms_per_step'], {k:round(x,3) for k,x in bk.items() if 'bf3' in k}, 'frac', d['roofline'].get('frac'))
except Exception as e: print(v,'ERR',e)
P
done
