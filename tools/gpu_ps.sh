#!/bin/bash
# persistent-path check: parity tests of the three programs, stage timelines, c2 bench (twice)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ps; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "persistent or driver_timed" 2>&1 | tail -3
for p in dp enc flow; do timeout 120 python tools/ps_trace.py $p 2>&1 | tail -${PS_TAIL:-4}; done
for k in 1 2; do
  BENCH_SKIP_FINITE_CHECK=1 timeout 200 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
print(d['ms_per_step'], {k:round(v,4) for k,v in bk.items() if 'persist' in k})"
done
