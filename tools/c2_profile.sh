set -e
# rocprofv3 kernel summary of the c2-only bench (headline workload, graph replay); copies the csv under gpurun_out/ as r2_c2_only_*
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/c2_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_prof -o trace -- python $R/bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 > $O/${TAG}_c2_only_bench_under_rocprof.json.txt 2> $O/c2_prof.err
cp $(find $O/c2_prof -name '*kernel_stats.csv' | head -1) $O/${TAG}_c2_only_bench_rocprofv3_kernel_stats.csv
rm -rf $O/c2_prof
head -40 $O/${TAG}_c2_only_bench_rocprofv3_kernel_stats.csv | cut -c1-170
