#!/bin/bash
# BERT encoder: FFN second matrix K-sliced (default) vs single launch (VITS_BERT_KSLICE=0): parity, then ms per encode at sentence sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/bert2; mkdir -p $O
timeout 600 python -m pytest tests/test_stts_hip_parity.py -m gpu -x -q -k "bert" 2>&1 | tail -4 | tee $O/parity.txt
for rep in 1 2; do
  for T in 12 24 60; do
    for k in 1 0; do
      echo -n "KSLICE=$k " ; VITS_BERT_KSLICE=$k timeout 120 python tools/bert_profile.py $T 400 2>&1 | tail -1
    done
  done
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o trace -- python $R/tools/bert_profile.py 12 200 > $R/$O/run.txt 2>&1
f=$(find $R/$O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/bert_kernel_stats.csv
rm -rf $R/$O/prof
