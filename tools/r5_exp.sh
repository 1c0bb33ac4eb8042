#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/exp1; mkdir -p $O; cd $R
SH="1,768,192,3520,3,1;8,256,256,564,3,1"
for v in expC expE expF expG expH; do
  echo "== $v"; BT_LIB=$R/vosk_tts_amd/csrc/libvits_mi355_$v.so VITS_SP=2 timeout 300 python tools/bt_conv.py "$SH" 2>&1 | grep -E "workgroups on|conv dbg" | sed -e 's/; start.*//' -e 's/: last launch.*launches =/ ->/'
done > $O/exp.txt 2>&1
cat $O/exp.txt
