#!/bin/bash
# Round evidence: the default bench line and the rocprofv3 kernel-trace summary of THE SAME command, plus the
# per-workload bench lines.  Run on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh r1'
# Outputs land in gpurun_out/<tag>_* ; copy the ones to be judged into profiles/.
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_default_bench.json.txt 2> $O/${TAG}_default_bench.err
for w in c3 c4 c5; do python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_${w}_bench.json.txt 2>> $O/${TAG}_default_bench.err; done
python bench.py --workload m2 --steps 30 --warmup 3 > $O/${TAG}_m2_bench.json.txt 2>> $O/${TAG}_default_bench.err
python bench.py --workload m3 --steps 5 --warmup 2 --cpu-seconds 5 > $O/${TAG}_m3_bench.json.txt 2>> $O/${TAG}_default_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o trace -- python $R/bench.py > $O/${TAG}_default_bench_under_rocprof.json.txt 2>> $O/${TAG}_default_bench.err
cp $(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1) $O/${TAG}_default_bench_rocprofv3_kernel_stats.csv
rm -rf $O/${TAG}_prof
ls -la $O | grep ${TAG}_
