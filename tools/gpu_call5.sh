#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/call5; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/default.json 2> $O/default.err; echo "default rc=$?"; tail -c 400 $O/default.err
timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rc=$?"; tail -c 600 $O/gpus2.err
python - <<'P'
import json
for f in ('default','gpus2'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/call5/{f}.json').read().splitlines() if l.startswith('{')][-1])
        print(f, d['n_gpus'], d['ms_per_step'], d['value'], 'ranks_seen', d['ranks_seen'], d['launched_by'], d['rank_ms'])
        print('  batch256', d['batch256_sharded'])
        print('  mds', d['multi_device_synth'])
        print('  host', d['host_api'] and d['host_api']['free_running']['ms_median'], 'batch32', d['batch32'] and d['batch32']['ms_per_step'], 'bf16x3', d['batch32_bf16x3'] and d['batch32_bf16x3']['ms_per_step'], 'm2', d['multistream'] and d['multistream']['ms_per_step'])
        print('  roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'fwd frac', d['roofline']['forward']['frac'])
        print('  cpu', d['cpu_baseline'] and (d['cpu_baseline']['value'], d['cpu_baseline']['cores']))
    except Exception as e: print(f,'ERR',e)
P
