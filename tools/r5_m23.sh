#!/bin/bash
# after settle_gc() in bench.py: m2 against the number of timed steps again, then the default line and m2 / m3 lines for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/final2; mkdir -p $O
run() { python bench.py --workload m2 --no-cpu-baseline --no-host-api "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "--steps 10 --warmup 3" "--steps 50 --warmup 5" "--steps 200 --warmup 5"; do
  echo "$cfg: $(run $cfg) ms"
done | tee $O/m2_steps_after.txt
for w in m2 m3; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-api > $O/r5_${w}_bench.json.txt 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py > $O/r5_default_bench.json.txt 2> $O/bench_default.err; echo "bench rc=$?"
python tools/bench_summary.py $O/r5_default_bench.json.txt | head -30
