#!/bin/bash
# round-2 GPU run 1: full GPU test suite + K-split wave-count sweep on c2 + the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r2_t1.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t1.log
tail -15 $O/r2_t1.log
for nw in 16 0; do
  VITS_KS_WAVES=$nw timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/r2_c2_nw$nw.json 2> $O/r2_c2_nw$nw.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c2_nw$nw.json").read().strip().splitlines()[-1])
    print("nw=$nw ms/step", d["ms_per_step"], "launches", d.get("launches_per_forward"))
    print("  by_op", d["roofline"]["by_op_ms_per_forward"])
except Exception as e:
    print("nw=$nw failed", e); print(open("$O/r2_c2_nw$nw.err").read()[-2000:])
PY
done
timeout 600 python bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$O/r2_bench_default.json").read().strip().splitlines()[-1])
    print("default: ms/step", d["ms_per_step"], "x_rt", d["x_realtime"], "timed_s", d.get("timed_region_s"))
    print("host_api", json.dumps(d.get("host_api")))
    print("batch32 ms", d["batch32"]["ms_per_step"] if d.get("batch32") else None, "m2", d["multistream"]["ms_per_step"] if d.get("multistream") else None)
except Exception as e:
    print("default failed", e); print(open("$O/r2_bench_default.err").read()[-3000:])
PY
