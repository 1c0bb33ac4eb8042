R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "persistent or c2_single or timeout or fast_path or mid_size or free_running" 2>&1 | tail -5 | tee $O/tests512.txt
for w in u110 u130 u150 u170 u250 u400; do for c in "" "VITS_PS_MAX_T=256"; do
echo "[$c] $w: $(env $c timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-api --no-extras --no-batch32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
o=r['by_op_ms_per_forward']
dec=sum(v for k,v in o.items() if k.startswith('dec.') or k=='istft_pqmf')
print(d['ms_per_step'], 'decoder', round(dec,3), 'rest', round(d['ms_per_step']-dec,3), {k:round(v,3) for k,v in o.items() if 'persist' in k})")"
done; done 2>&1 | tee -a $O/persist_512.txt
