#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stts_hip_parity.py tests/test_host_api.py -m gpu -q -x --timeout 600 > $O/r2_t8.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t8.log
tail -5 $O/r2_t8.log
timeout 300 python bench.py --workload m2 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m2 ms/step', d['ms_per_step'], 'x_rt', d['x_realtime'])"
timeout 300 python bench.py --workload m3 --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m3 ms/step', d['ms_per_step'], 'x_rt', d['x_realtime'])"
