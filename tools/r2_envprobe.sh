#!/bin/bash
# does where the runtime keeps kernel arguments / how it replays graphs change the per-launch floor of the c2 forward?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() {
  env "$@" VITS_KS_WAVES=16 timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step', d['ms_per_step'], 'eager sum', d['roofline']['forward']['sum_kernel_ms_eager'])"
}
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GPU_MAX_HW_QUEUES=1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/launchprobe tools/launchprobe.hip && /tmp/launchprobe | head -4
HIP_FORCE_DEV_KERNARG=0 /tmp/launchprobe | head -4
