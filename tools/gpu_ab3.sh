#!/bin/bash
# A/B of library variants (tools/bt/bt_*.so) on the c3 split-bf16 bench: ms_per_step and per-forward device time of the bf3 kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab3; mkdir -p $O; cd $R
for so in tools/bt/bt_*.so; do
  v=$(basename $so .so)
  BENCH_SKIP_FINITE_CHECK=1 VITS_MI355_LIB=$R/$so VITS_BF3_PC=0 timeout 300 python bench.py --workload c3 --precision bf16x3 --no-cpu-baseline --no-host-api --steps 10 --warmup 3 > $O/$v.json 2> $O/$v.err || echo "$v failed: $(tail -2 $O/$v.err)"
  python - $v <<'P'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/ab3/{v}.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
    print(v, d['ms_per_step'], {k:round(x,3) for k,x in bk.items() if 'bf3' in k}, 'frac', d['roofline'].get('frac'))
except Exception as e: print(v,'ERR',e)
P
done
