#!/bin/bash
# experiment: conv_wp at batch size (c3) instead of the big-tile kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
VITS_WP_BIG=1 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 900 -k "c3_full or poisoned or ragged" > $O/r2_t13.log 2>&1; echo "pytest rc=$?"
tail -5 $O/r2_t13.log
for cfg in "X=1" "VITS_WP_BIG=1"; do
  env $cfg timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-host-api --steps 10 > $O/r2_c3_$cfg.json 2> $O/r2_c3.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c3_$cfg.json").read().strip().splitlines()[-1])
    print("[$cfg] c3 ms/step", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
    print("   by_kernel", d["roofline"]["by_kernel_ms_per_forward"])
except Exception as e:
    print("c3 failed", e); print(open("$O/r2_c3.err").read()[-2000:])
PY
done
