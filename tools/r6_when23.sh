#!/bin/bash
# a single-utterance call takes the persistent programs when at most VITS_PERSIST_WHEN other host calls are in flight as it starts (-1: whenever the token is free)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for n in -1 0 1 2; do
  BENCH_THREADS=${BENCH_THREADS:-1,2,3,4,8,16} VITS_PERSIST_WHEN=$n timeout 600 python bench.py --no-batch32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['host_api']['concurrent']
print('when=$n rep=$rep |', ' | '.join('%dthr %.0f req/s p50 %.2f p90 %.2f (persist %s/%s)' % (x['threads'], x['requests_per_s'], x['ms_p50'], x['ms_p90'], x.get('persistent_launches'), x.get('engine_calls')) for x in c['coalesced']))"
done; done
