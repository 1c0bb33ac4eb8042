#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bert1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "bert or Bert" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/tests.log
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from vosk_tts_amd import weights as W
print(bench.bert_voice_leg(W.default_hparams(), 0))
PY
