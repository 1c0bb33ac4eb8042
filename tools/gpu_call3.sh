#!/bin/bash
# persistent enc / sdp / flow: c2 by mask + full test suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/call3; mkdir -p $O; cd $R
for v in 0 1 3 7; do
  VITS_PERSIST=$v timeout 300 python bench.py --no-batch32 --no-cpu-baseline --steps 50 > $O/c2_m$v.json 2> $O/c2_m$v.err; echo "bench c2 mask $v rc=$?"
done
python - <<'P'
import json
for v in (0,1,3,7):
    try:
        d=json.loads(open(f'gpurun_out/call3/c2_m{v}.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
        print(v, d['ms_per_step'], d['launches_per_forward'], d['host_api']['free_running']['ms_median'], d['host_api']['pinned']['ms_median'], {k:v for k,v in bk.items() if 'persist' in k}, {k:v for k,v in d['roofline']['by_op_ms_per_forward'].items() if 'persist' in k})
    except Exception as e: print(v,'ERR',e)
P
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
