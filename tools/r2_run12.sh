#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/m2_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m2_prof -o trace -- python $R/bench.py --workload m2 --no-cpu-baseline --steps 30 > $O/r2_m2_under_rocprof.json.txt 2> $O/m2_prof.err
cp $(find $O/m2_prof -name '*kernel_stats.csv' | head -1) $O/r2_m2_bench_rocprofv3_kernel_stats.csv
rm -rf $O/m2_prof
head -30 $O/r2_m2_bench_rocprofv3_kernel_stats.csv | cut -c1-150
