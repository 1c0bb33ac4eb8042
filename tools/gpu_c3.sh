#!/bin/bash
# c3 both precisions (per-kernel table) + the parity tests that cover batch launches
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "${C3_TESTS:-bf16x3 or batch or ragged}" 2>&1 | tail -3
for prec in bf16x3 f32; do
  for k in 1 2; do
  BENCH_SKIP_FINITE_CHECK=1 VITS_BF3_PC=${BF3_PC:-0} timeout 300 python bench.py --workload c3 --precision $prec --no-cpu-baseline --no-host-api --steps 10 --warmup 3 > $O/$prec.json 2> $O/$prec.err || echo "$prec failed: $(tail -2 $O/$prec.err)"
  python - $prec <<'P'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/c3/{v}.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
    top=sorted(bk.items(), key=lambda kv:-kv[1])[:6]
    print(v, d['ms_per_step'], {k:round(x,3) for k,x in top}, 'frac', d['roofline'].get('frac'))
except Exception as e: print(v,'ERR',e)
P
  done
done
