#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in 1 0 1 0; do for prec in f32 bf16x3; do
  BENCH_SKIP_FINITE_CHECK=1 VITS_PAIR_MTILES=$m timeout 300 python bench.py --workload c3 --precision $prec --no-cpu-baseline --no-host-api --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']; top=sorted(bk.items(), key=lambda kv:-kv[1])[:2]
print('pair', $m, '$prec', d['ms_per_step'], {k:round(x,3) for k,x in top})"
done; done
