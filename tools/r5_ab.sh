#!/bin/bash
# end-to-end A/B of an environment switch on the bench workloads:  tools/r5_ab.sh <outdir> "<env a>" "<env b>" ... ; workloads in $WL
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
WL=${WL:-"c3 s8 s16 c5 u300"}
for e in "$@"; do
  for w in $WL; do
    ms=$(env $e timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-api --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "[$e] $w: $ms ms"
  done
done | tee $O/ab.txt
