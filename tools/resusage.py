"""Table of registers / spills / occupancy per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log)."""
import re, sys, subprocess
log = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for line in log.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    try:
        dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dn = name
    dn = dn.replace("void ", "").split("(")[0]
    if pat and pat not in dn:
        continue
    print(f"{dn:60s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>4} SGPR {r.get('TotalSGPRs','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
