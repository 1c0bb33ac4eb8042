#!/bin/bash
# A/B of the XCD ownership of single-utterance conv launches (VITS_XCD_MAP: 0 legacy order, 1 per-launch choice, 2 column-major, 3 M-major)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/xcd; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "stts_streamed or stream or decoder or full" 2>&1 | tail -3
for m in 0 1 2 3 0 1; do
  BENCH_SKIP_FINITE_CHECK=1 VITS_XCD_MAP=$m timeout 200 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/m$m.json 2> $O/m$m.err || echo "m$m failed: $(tail -2 $O/m$m.err)"
  python - $m <<'P'
import json,sys
m=sys.argv[1]
d=json.loads(open(f'gpurun_out/xcd/m{m}.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel_ms_per_forward']
print('map',m, d['ms_per_step'], {k:round(v,4) for k,v in bk.items() if 'conv_wp' in k or '_ks_' in k})
P
done
