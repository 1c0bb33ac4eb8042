#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -k "lds_staged or stages_full or plain_generator or c2_single or streaming_chunks or tail or hifigan" > $O/r2_t5.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t5.log
tail -25 $O/r2_t5.log
for m in 1 2; do
VITS_CONV_LS=$m VITS_KS_WAVES=16 timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/r2_c2_v5_$m.json 2> $O/r2_c2_v5_$m.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r2_c2_v5_$m.json").read().strip().splitlines()[-1])
    print("conv_ls=$m ms/step", d["ms_per_step"], "launches", d.get("launches_per_forward"))
    print("  by_op", {k:v for k,v in d["roofline"]["by_op_ms_per_forward"].items() if k.startswith("dec.")})
    print("  by_kernel", d["roofline"]["by_kernel_ms_per_forward"])
except Exception as e:
    print("failed", e); print(open("$O/r2_c2_v5_$m.err").read()[-2000:])
PY
done
