#!/bin/bash
# conv_wp_kernel: parity tests, single-launch timings (tools/convdbg.py) and the c2 bench with / without it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "wave_pipelined or fast_path or c2_single or poisoned" > $O/r2_t9.log 2>&1; echo "pytest rc=$?"
tail -12 $O/r2_t9.log
for cfg in "VITS_CONV_WP=2"; do
  echo "=== $cfg"
  env $cfg VITS_CONV_DBG=20 timeout 300 python tools/convdbg.py decoder 2>&1 | grep -E "conv dbg|wave" | cut -c1-200 | grep -A9 -E "T=2400 K=11"
done
for wp in 1 0; do
  VITS_CONV_WP=$wp timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/r2_c2_wp$wp.json 2> $O/r2_c2_wp$wp.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c2_wp$wp.json").read().strip().splitlines()[-1])
    print("wp=$wp ms/step", d["ms_per_step"], "launches", d.get("launches_per_forward"))
    print("  by_op", {k:v for k,v in d["roofline"]["by_op_ms_per_forward"].items() if k.startswith("dec")})
except Exception as e:
    print("wp=$wp failed", e); print(open("$O/r2_c2_wp$wp.err").read()[-2000:])
PY
done
