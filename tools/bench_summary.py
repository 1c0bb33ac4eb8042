"""One-screen summary of a bench.py JSON line (file argument): the figures the round's review asks for."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2 ms_per_step", d["ms_per_step"], "x_realtime", d.get("x_realtime"), "graph nodes", d.get("launches_per_forward"))
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("kernel", "achieved", "frac", "traffic", "kernel_ms_per_forward", "avg_launch_us", "kernel_launches_per_forward")})
print("  conv_kernel", r.get("conv_kernel"))
print("  forward", r.get("forward"))
print("  by_op", r.get("by_op_ms_per_forward"))
if d.get("streaming"):
    print("streaming", {k: v for k, v in d["streaming"].items() if k not in ("note", "workload")})
h = d.get("host_api") or {}
for k in ("free_running", "pinned"):
    if k in h:
        print("host_api", k, {x: h[k][x] for x in ("ms_median", "ms_p90", "x_realtime")})
if "concurrent" in h:
    c = h["concurrent"]
    if "error" in c:
        print("concurrent ERROR", c["error"])
    else:
        for leg in c["coalesced"]:
            print("concurrent", leg)
        print("uncoalesced 16", c["uncoalesced_16_threads"])
        print("speedup 16 over 1:", c["speedup_16_threads_over_1"], "tokens", c["text_tokens"])
for k in ("batch32", "batch32_bf16x3"):
    if d.get(k):
        b = d[k]
        print(k, b["ms_per_step"], "x_rt", b.get("x_realtime"), "kernel", b["roofline"].get("kernel"), "frac", b["roofline"].get("frac"), "forward", b["roofline"].get("forward"))
if d.get("multistream"):
    print("multistream", d["multistream"]["ms_per_step"], d["multistream"]["x_realtime"])
if d.get("batch256_sharded"):
    print("batch256", d["batch256_sharded"]["ms_per_step"], d["batch256_sharded"]["x_realtime"])
if d.get("cpu_baseline"):
    print("cpu_baseline x_rt", d["cpu_baseline"]["x_realtime"], "cores", d["cpu_baseline"]["cores"])
