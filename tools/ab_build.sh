#!/bin/bash
# Builds alternative product libraries into tools/bt/ (git-ignored, travels to the GPU box) for A/B runs on one lease:
#   tools/ab_build.sh base                       working tree as is            -> tools/bt/bt_base.so
#   tools/ab_build.sh exp:-DSOME_SWITCH          working tree with extra flags -> tools/bt/bt_exp.so
#   tools/ab_build.sh old@<git ref>              sources of a git ref          -> tools/bt/bt_old.so
# then on the GPU box:  VITS_MI355_LIB=tools/bt/bt_exp.so python bench.py ...   (tools/gpu_lease.sh libs)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/bt $R/gpurun_out
for spec in "$@"; do
  name=${spec%%[:@]*}; flags=""; ref=""
  case "$spec" in *:*) flags=${spec#*:};; *@*) ref=${spec#*@};; esac
  src=$R/vosk_tts_amd/csrc
  if [ -n "$ref" ]; then
    tmp=$(mktemp -d -p $R/gpurun_out)
    mkdir -p $tmp/vosk_tts_amd/csrc $tmp/include
    for f in $(git -C $R ls-tree --name-only $ref vosk_tts_amd/csrc/ include/ | grep -E '\.(hip|h)$'); do git -C $R show $ref:$f > $tmp/$f; done
    src=$tmp/vosk_tts_amd/csrc
  fi
  ( cd $src && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $flags -shared -o $R/tools/bt/bt_$name.so engine.hip && echo built tools/bt/bt_$name.so ) &
done
wait
