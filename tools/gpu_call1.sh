#!/bin/bash
# round 3, GPU call 1: LL-exchange probe + A/B of the fragment-wide STORE epilogue in conv_bf3_kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/call1; mkdir -p $O; cd $R
timeout 120 tools/llprobe > $O/llprobe.txt 2>&1; echo "llprobe rc=$?"
timeout 60 tools/xcdbarrierprobe > $O/xcdbarrierprobe.txt 2>&1; echo "xcdbarrier rc=$?"
for v in base epi; do
  VITS_MI355_LIB=$R/tools/bt/bt_$v.so timeout 300 python bench.py --workload c3 --precision bf16x3 --no-cpu-baseline --no-host-api > $O/c3_bf16x3_$v.json 2> $O/c3_bf16x3_$v.err; echo "bench c3 bf16x3 $v rc=$?"
done
VITS_MI355_LIB=$R/tools/bt/bt_epi.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf16x3 or c3_full or epilogue" > $O/pytest_epi.log 2>&1; echo "pytest epi rc=$?"; tail -3 $O/pytest_epi.log
cat $O/llprobe.txt
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/call1/c3_bf16x3_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
P
