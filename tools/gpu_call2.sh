#!/bin/bash
# persistent SDP: parity tests + c2 A/B + rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/call2; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_host_api.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in on off; do
  if [ $v == off ]; then export VITS_NO_PERSIST=1; else unset VITS_NO_PERSIST; fi
  timeout 300 python bench.py --no-batch32 --no-cpu-baseline --steps 50 > $O/c2_$v.json 2> $O/c2_$v.err; echo "bench c2 $v rc=$?"
done
unset VITS_NO_PERSIST
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/c2_prof.json 2> $O/prof.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/c2_kernel_stats.csv; rm -rf $O/prof
cd $R
python - <<'P'
import json,csv
for v in ('on','off'):
    try:
        d=json.loads(open(f'gpurun_out/call2/c2_{v}.json').read().strip().splitlines()[-1]); print(v, d['ms_per_step'], d['launches_per_forward'], d['host_api']['free_running']['ms_median'], d['host_api']['pinned']['ms_median'])
    except Exception as e: print(v,'ERR',e)
rows=list(csv.DictReader(open('gpurun_out/call2/c2_kernel_stats.csv')))
for r in rows[:12]: print(r['Calls'], r['AverageNs'], r['Name'][:80])
P
