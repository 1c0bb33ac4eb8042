#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -k "stages or epilogue or wn_folded or c2_single or fast_path or poisoned or free_running or c3_full or long_form" > $O/r2_t4.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t4.log
tail -25 $O/r2_t4.log
VITS_KS_WAVES=16 timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/r2_c2_v4.json 2> $O/r2_c2_v4.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c2_v4.json").read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], "launches", d.get("launches_per_forward"))
    print("  by_op", d["roofline"]["by_op_ms_per_forward"])
except Exception as e:
    print("failed", e); print(open("$O/r2_c2_v4.err").read()[-2000:])
PY
