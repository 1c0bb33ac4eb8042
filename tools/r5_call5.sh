#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bt1; mkdir -p $O; cd $R
SH="1,192,768,3520,3,1;8,256,256,564,3,1;8,256,256,564,11,5;1,192,384,10560,5,1;1,192,576,10560,1,1"
VITS_KS_THRESHOLD=1 VITS_CONV_WP=1 timeout 600 python tools/bt_conv.py "$SH" > $O/bt_64x64.txt 2>&1
cat $O/bt_64x64.txt
