#!/bin/bash
# serving throughput (bench.py host_api.concurrent): persistent programs whenever the token is free (VITS_PERSIST_WHEN=0) against only for calls that start alone (=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_when; mkdir -p $O; cd $R
for rep in 1 2; do
for n in 0 1; do
  VITS_PERSIST_WHEN=$n timeout 600 python bench.py --no-batch32 --no-cpu-baseline 2>$O/err_$n.txt | tail -1 > $O/line_$n.json
  python - <<P
import json
d=json.loads(open("$O/line_$n.json").read())
h=d["host_api"]; c=h["concurrent"]
print("when=$n rep=$rep c2 %.4f ms | free-running host call %.4f ms |" % (d["ms_per_step"], h["free_running"]["ms_median"]),
      " | ".join("%dthr %.0f req/s p50 %.2f p90 %.2f ms (persist launches %s, calls %s)" % (x["threads"], x["requests_per_s"], x["ms_p50"], x["ms_p90"], x.get("persistent_launches"), x.get("engine_calls")) for x in c["coalesced"] + [c["uncoalesced_16_threads"]]),
      "| bert_voice %.3f ms" % h["bert_voice"]["ms_median"] if h.get("bert_voice") else "")
P
done; done | tee $O/when.txt
