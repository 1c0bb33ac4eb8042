#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for prec in bf16x3; do
  timeout 300 python bench.py --workload c3 --precision $prec --no-cpu-baseline --no-host-api --steps 10 > $O/r2_c3_$prec.json 2> $O/r2_c3.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c3_$prec.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("[$prec] c3 ms/step", d["ms_per_step"], d["dtype"], r["kernel"], "achieved", r["achieved"], "peak", r["peak"], "frac", r["frac"], "avg_us", r["avg_launch_us"])
    print("   by_kernel", {k:v for k,v in list(r["by_kernel_ms_per_forward"].items())[:8]})
except Exception as e:
    print("c3 failed", e); print(open("$O/r2_c3.err").read()[-2000:])
PY
done
