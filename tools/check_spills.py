#!/usr/bin/env python3
"""Read register/scratch metadata of every gfx950 kernel in the built objects (no GPU needed).

A single spilled VGPR in the persistent NTT kernels costs ~40 % (its reload is an s_waitcnt vmcnt(0) queued
behind the HBM prefetch), so tests/test_abi.py asserts the headline kernels are spill-free.
usage: tools/check_spills.py [obj ...]   -> one line per kernel: name vgprs spills scratch lds"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                              stderr=subprocess.DEVNULL)
        txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    # amdhsa.kernels is a YAML list; every kernel entry starts with "  - .<first key>:" (keys are sorted, so
    # .group_segment_fixed_size precedes .name inside one entry)
    out, cur = [], None
    in_kernels = False
    for line in txt.splitlines():
        if line.strip().startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if not in_kernels:
            continue
        if re.match(r"^\S", line) and not line.startswith(" "):
            in_kernels = line.strip().startswith("amdhsa.kernels")
        m = re.match(r"^(\s*)(-\s+)?\.(\w+):\s*(.*)$", line)
        if not m: continue
        indent, dash, k, v = len(m.group(1)), m.group(2), m.group(3), m.group(4).strip()
        if dash and indent <= 2:
            cur = {}; out.append(cur)
        if cur is None or indent > 4: continue
        if k == "name": cur["name"] = v
        elif k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                   "group_segment_fixed_size", "agpr_count", "sgpr_count"):
            try: cur[k] = int(v)
            except ValueError: pass
    return [k for k in out if "name" in k and "vgpr_count" in k]


def all_kernels(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd", "csrc", "inst_*.o")))
    res = []
    for o in objs:
        for k in kernels_of(o):
            k["obj"] = os.path.basename(o); res.append(k)
    return res


if __name__ == "__main__":
    for k in all_kernels(sys.argv[1:] or None):
        print("%-90s vgpr %3d spill %3d scratch %4d lds %6d" % (k["name"][:90], k.get("vgpr_count", -1), k.get("vgpr_spill_count", -1),
                                                               k.get("private_segment_fixed_size", -1), k.get("group_segment_fixed_size", -1)))
