#!/bin/bash
# serving throughput (bench.py host_api.concurrent) against the number of callers that may run persistent programs on the device at once
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_owners; mkdir -p $O; cd $R
for rep in 1 2; do
for n in ${OWNERS:-1 2 3}; do
  VITS_PERSIST_OWNERS=$n timeout 600 python bench.py --no-batch32 --no-cpu-baseline 2>$O/err_$n.txt | tail -1 > $O/line_$n.json
  python - <<P
import json
d=json.loads(open("$O/line_$n.json").read())
h=d["host_api"]; c=h["concurrent"]
st=h.get("persist_state") or {}
print("owners=$n rep=$rep c2 %.4f ms | free-running host call %.4f ms |" % (d["ms_per_step"], h["free_running"]["ms_median"]),
      " | ".join("%dthr %.0f req/s p50 %.2f ms (persist launches %s, calls %s)" % (x["threads"], x["requests_per_s"], x["ms_p50"], x.get("persistent_launches"), x.get("engine_calls")) for x in c["coalesced"] + [c["uncoalesced_16_threads"]]))
P
done; done | tee $O/owners.txt
