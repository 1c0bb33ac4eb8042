#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/call4; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "persistent" 2>&1 | tail -3
for v in 0 7; do
  VITS_PERSIST=$v timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/c2_m$v.json 2> $O/c2_m$v.err; echo "bench c2 mask $v rc=$?"
done
python - <<'P'
import json
for v in (0,7):
    try:
        d=json.loads(open(f'gpurun_out/call4/c2_m{v}.json').read().strip().splitlines()[-1])
        print(v, d['ms_per_step'], d['launches_per_forward'], {k:v for k,v in d['roofline']['by_op_ms_per_forward'].items() if 'persist' in k})
    except Exception as e: print(v,'ERR',e)
P
for p in dp enc flow; do python tools/ps_trace.py $p 2>&1 | tail -${TRACE_TAIL:-12}; done
