#!/bin/bash
# serving throughput against the coalescer's in-flight limit under the "programs only for calls that start alone" rule
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_inflight; mkdir -p $O; cd $R
for n in ${INFLIGHT:-2 3 4 6 8 12}; do
  VITS_COALESCE_INFLIGHT=$n timeout 600 python bench.py --no-batch32 --no-cpu-baseline 2>$O/err_$n.txt | tail -1 > $O/line_$n.json
  python - <<P
import json
d=json.loads(open("$O/line_$n.json").read())
h=d["host_api"]; c=h["concurrent"]
print("inflight=$n |", " | ".join("%dthr %.0f req/s p50 %.2f p90 %.2f ms (calls %s, mean batch %s)" % (x["threads"], x["requests_per_s"], x["ms_p50"], x["ms_p90"], x.get("engine_calls"), x.get("mean_batch")) for x in c["coalesced"]))
P
done | tee $O/inflight.txt
