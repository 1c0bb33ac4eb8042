#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/full1; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -8 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/default_bench.json.txt 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 400 $O/bench_default.err
python tools/bench_summary.py $O/default_bench.json.txt
for w in m2 m3; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-api > $O/${w}_bench.json.txt 2> $O/bench_$w.err; echo "bench $w rc=$?"; python -c "
import json,sys; d=json.loads(open('$O/${w}_bench.json.txt').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['x_realtime'])"; done
