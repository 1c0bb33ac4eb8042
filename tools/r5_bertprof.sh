#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bertprof; mkdir -p $O; cd $R
python tools/bert_profile.py 12 200; python tools/bert_profile.py 30 200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/tools/bert_profile.py 12 200 > $O/run.txt 2>&1
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/bert_kernel_stats.csv; rm -rf $O/prof
head -12 $O/bert_kernel_stats.csv | cut -c1-150
