#!/bin/bash
# stts fast path + generalized conv_wp: parity tests, m2 / c2 bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_stts_hip_parity.py tests/test_hip_parity.py -m gpu -q -x --timeout 900 -k "stts or wave_pipelined or fast_path or c2_single or poisoned or stages" > $O/r2_t11.log 2>&1; echo "pytest rc=$?"
tail -8 $O/r2_t11.log
VITS_CONV_WP=2 timeout 600 python -m pytest tests/test_stts_hip_parity.py -m gpu -q -x --timeout 900 > $O/r2_t11b.log 2>&1; echo "pytest (wp everywhere) rc=$?"
tail -5 $O/r2_t11b.log
for w in m2 c2 m3; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-batch32 --no-host-api --steps 30 > $O/r2_$w.json 2> $O/r2_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/r2_$w.json").read().strip().splitlines()[-1])
    print("$w ms/step", d["ms_per_step"], "x_rt", d.get("x_realtime"))
except Exception as e:
    print("$w failed", e); print(open("$O/r2_$w.err").read()[-2000:])
PY
done
