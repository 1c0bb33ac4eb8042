set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/c2_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_prof -o trace -- python $R/bench.py --no-batch32 --no-cpu-baseline > $O/r1_c2_only_bench_under_rocprof.json.txt 2> $O/c2_prof.err
cp $(find $O/c2_prof -name '*kernel_stats.csv' | head -1) $O/r1_c2_only_bench_rocprofv3_kernel_stats.csv
rm -rf $O/c2_prof
head -12 $O/r1_c2_only_bench_rocprofv3_kernel_stats.csv | cut -c1-150
