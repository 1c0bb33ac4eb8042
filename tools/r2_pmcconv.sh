#!/bin/bash
# PMC counters of single decoder-shaped conv launches (K-split kernel, 16 waves): where do the operands come from?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcconv; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/convdbg.py decoder"
export VITS_KS_WAVES=16 VITS_CONV_LS=1 VITS_CONV_DBG=6 CONVDBG_PLAIN=1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O -o tcc -- $CMD > /dev/null 2>&1 || echo pass1 failed
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O -o sq -- $CMD > /dev/null 2>&1 || echo pass2 failed
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O -o tcp -- $CMD > /dev/null 2>&1 || echo pass3 failed
python - <<PY
import csv, glob, collections
for tag in ("tcc","sq","tcp"):
    fs = glob.glob("$O/**/%s_counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no csv"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for (kn, cn), (n, v) in agg.items():
        if "ks_kernel" in kn or "conv16" in kn: print(tag, kn, cn, "launches", n, "avg %.0f" % (v / n))
PY
