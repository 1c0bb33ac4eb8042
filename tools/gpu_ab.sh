#!/bin/bash
# A/B of library variants (tools/bt/bt_*.so) on the c2 bench: ms_per_step and the per-forward device time of selected kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for so in tools/bt/bt_*.so; do
  v=$(basename $so .so)
  BENCH_SKIP_FINITE_CHECK=1 VITS_MI355_LIB=$R/$so timeout 200 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 ${BENCH_EXTRA} > $O/$v.json 2> $O/$v.err || echo "$v failed: $(tail -2 $O/$v.err)"
done
python - <<'P'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/ab/bt_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
        print(os.path.basename(f), d['ms_per_step'], {k:v for k,v in bk.items() if 'persist' in k})
    except Exception as e: print(f,'ERR',e)
P
