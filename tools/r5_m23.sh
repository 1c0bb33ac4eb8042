#!/bin/bash
# m3 (32 ragged StableTTS utterances): repeatability on ONE box (127.6 - 140.2 ms were seen on four boxes of round 5 while c3 / c4 held +-0.3 %)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/final2; mkdir -p $O
run() { python bench.py --workload m3 --no-cpu-baseline --no-host-api "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "--steps 10 --warmup 3" "--steps 10 --warmup 3" "--steps 30 --warmup 3" "--steps 10 --warmup 3"; do
  echo "$cfg: $(run $cfg) ms"
done | tee $O/m3_repeat.txt
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 | tee -a $O/m3_repeat.txt
