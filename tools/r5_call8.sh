#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bt3; mkdir -p $O; cd $R
SH="8,256,256,564,3,1;1,768,192,3520,3,1;1,192,192,10560,1,1"
BT_LIB=$R/vosk_tts_amd/csrc/libvits_mi355_exp.so VITS_SP=2 timeout 600 python tools/bt_conv.py "$SH" > $O/bt_sp_samew.txt 2>&1
cat $O/bt_sp_samew.txt
