"""Where does hipcc wait for EVERYTHING in flight inside a loop?  Compiles engine.hip to gfx950 assembly (no GPU needed) and lists, per
kernel, the `s_waitcnt vmcnt(0)` that sit inside loops and are followed by register moves -- the signature of a loop-carried value
whose load (or whose producer's store, both count in vmcnt on gfx9) is still in flight at the back edge.  Round 3 found two of
these by hand (persist_kernel: -10 %; the split-bf16 chunk loop: needs registers it does not have); this is the first look for the
other kernels.       python tools/isa_waits.py [substring of a kernel name]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "engine_gfx950.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-w", "-S", "-o", out, "engine.hip"],
                      cwd=os.path.join(root, "vosk_tts_amd", "csrc"))
t = open(out).read()
names = re.findall(r"^(_Z\S+|\w+):\s*; @", t, flags=re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1] if len(sys.argv) > 1 else ""
for m, d in zip(names, dem):
    if want not in d:
        continue
    i = t.index("\n" + m + ":"); j = t.index(".Lfunc_end", i)
    raw = t[i:j].split("\n")
    hits = []
    in_loop = False
    for k, l in enumerate(raw):
        s = l.strip()
        if s.startswith(".LBB"):
            in_loop = "in Loop" in l
        if in_loop and s.startswith("s_waitcnt") and "vmcnt(0)" in s:
            nxt = [x.strip() for x in raw[k + 1:k + 4]]
            if any(x.startswith("v_mov") for x in nxt):
                hits.append((k, sum(1 for x in raw[k + 1:k + 40] if x.strip().startswith("v_mov"))))
    if hits:
        print(f"{d.split('(')[0][:80]:80s} {len(hits)} in-loop vmcnt(0)+moves; moves behind each: {[h[1] for h in hits][:12]}")
