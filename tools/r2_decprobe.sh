#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() {
  env "$@" timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['roofline']['by_op_ms_per_forward']
print('$*', 'ms/step', d['ms_per_step'], {k:v for k,v in o.items() if k.startswith('dec.')})"
}
run VITS_KS_WAVES=16
run VITS_KS_WAVES=16 VITS_KS_SHAPE=12
run VITS_KS_WAVES=8 VITS_KS_SHAPE=12
run VITS_KS_WAVES=8
