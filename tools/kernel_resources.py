#!/usr/bin/env python3
"""Print per-kernel register/LDS usage of one instantiation file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-Iinclude", "-Iinclude/internal", "-Ioptimized-number-theoretic-transform-implementations_amd/csrc",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?)(?: \[-Rpass)", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
for r in rows:
    n = demangle(r["name"])
    if pat and pat not in n: continue
    n = re.sub(r"\(.*", "", n).replace("void ntt::", "")
    print("%-62s VGPR %4s spill %3s SGPR %4s scratch %5s LDS %7s occ %s" % (
        n, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("TotalSGPRs", r.get("SGPRs")), r.get("ScratchSize [bytes/lane]"),
        r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
