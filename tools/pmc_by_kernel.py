#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes PER KERNEL: for every kernel whose name contains PATTERN, the mean of each counter over its
full-size dispatches, all passes under ROOT merged into one table (columns = kernels), plus the ratios the block-kernel
analysis uses (per wave, per SIMD cycle).
usage: tools/pmc_by_kernel.py ROOT [PATTERN [COEFFICIENTS_PER_DISPATCH]]      (ROOT/<pass>/**/*counter_collection.csv, one pass per counter group)"""
import collections, csv, glob, re, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "fused_kernel"
coefs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # coefficients one dispatch processes (tools/sweep.py --bytes B: B / 8)


def short(name):
    m = re.search(r"(\w+)<ntt::(\w+(?:<\d+>)?), (\d+), (true|false)", name)
    if m: return "%s<%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), "inv" if m.group(4) == "true" else "fwd")
    return re.sub(r"\s+", " ", name.split("(")[0].replace("void ntt::", "").replace("ntt::", ""))[:60]


table = collections.defaultdict(dict)   # counter -> kernel -> mean
meta = {}
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    by_disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        d = by_disp[(short(r["Kernel_Name"]), r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["_grid"] = int(r["Grid_Size"]); d["_wg"] = int(r.get("Workgroup_Size", 0) or 0)
    kernels = sorted({k for k, _ in by_disp})
    for k in kernels:
        ds = [d for (kk, _), d in by_disp.items() if kk == k]
        names = [n for n in ds[0] if not n.startswith("_")]
        mx = max(d[names[0]] for d in ds)
        full = [d for d in ds if d[names[0]] > 0.5 * mx]   # drops warm-up / parity probes on a few polynomials
        meta[k] = (full[0]["_grid"], full[0]["_wg"], len(full))
        for n in names:
            table[n][k] = sum(d[n] for d in full) / len(full)
kernels = sorted(meta)
if not kernels:
    sys.exit("no kernel matching %r under %s" % (pat, root))
w = max(len(k) for k in kernels) + 2
print("%-28s" % "counter (mean per dispatch)" + "".join("%*s" % (w, k) for k in kernels))
print("%-28s" % "grid threads / wg / n" + "".join("%*s" % (w, "%d/%d/%d" % meta[k]) for k in kernels))
for n in sorted(table):
    print("%-28s" % n + "".join("%*s" % (w, ("%.4g" % table[n][k]) if k in table[n] else "-") for k in kernels))


def ratio(label, num, den, scale=1.0):
    if num in table and den in table:
        print("%-28s" % label + "".join("%*s" % (w, ("%.3f" % (scale * table[num][k] / table[den][k])) if k in table[num] and k in table[den] and table[den][k] else "-") for k in kernels))


print("# ratios (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)")
if coefs and "SQ_INSTS_VALU" in table:
    print("%-28s" % "VALU instr per coefficient" + "".join("%*s" % (w, "%.2f" % (64 * table["SQ_INSTS_VALU"][k] / coefs)) for k in kernels))
    if "SQ_INSTS_LDS" in table: print("%-28s" % "LDS instr per 16 coeff." + "".join("%*s" % (w, "%.2f" % (64 * 16 * table["SQ_INSTS_LDS"][k] / coefs)) for k in kernels))
    if "SQ_INSTS_SALU" in table: print("%-28s" % "SALU instr per 16 coeff." + "".join("%*s" % (w, "%.2f" % (64 * 16 * table["SQ_INSTS_SALU"][k] / coefs)) for k in kernels))
ratio("VALU instr per wave", "SQ_INSTS_VALU", "SQ_WAVES")
ratio("SALU instr per wave", "SQ_INSTS_SALU", "SQ_WAVES")
ratio("SMEM instr per wave", "SQ_INSTS_SMEM", "SQ_WAVES")
ratio("LDS instr per wave", "SQ_INSTS_LDS", "SQ_WAVES")
ratio("VMEM rd instr per wave", "SQ_INSTS_VMEM_RD", "SQ_WAVES")
ratio("VMEM wr instr per wave", "SQ_INSTS_VMEM_WR", "SQ_WAVES")
ratio("wait_any / wave_cycles", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
ratio("wait_inst_any / wave_cycles", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
ratio("wait_inst_lds / wave_cycles", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")
ratio("active_inst_any / wave_cyc", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
ratio("active_valu / wave_cycles", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
ratio("active_lds / wave_cycles", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES")
ratio("lds conflict / lds active", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
# VALU pipe busy: quad-cycles of VALU issue per SIMD over the kernel's cycles: 1024 SIMDs, GUI_ACTIVE / 8 cycles per dispatch
if "SQ_ACTIVE_INST_VALU" in table and "GRBM_GUI_ACTIVE" in table:
    print("%-28s" % "VALU busy (4*act/1024/cyc)" + "".join("%*s" % (w, "%.3f" % (4 * table["SQ_ACTIVE_INST_VALU"][k] / 1024 / (table["GRBM_GUI_ACTIVE"][k] / 8)) if k in table["SQ_ACTIVE_INST_VALU"] and k in table["GRBM_GUI_ACTIVE"] else "-") for k in kernels))
if "SQ_LDS_IDX_ACTIVE" in table and "GRBM_GUI_ACTIVE" in table:
    print("%-28s" % "LDS busy (idx_act/256/cyc)" + "".join("%*s" % (w, "%.3f" % (table["SQ_LDS_IDX_ACTIVE"][k] / 256 / (table["GRBM_GUI_ACTIVE"][k] / 8)) if k in table["SQ_LDS_IDX_ACTIVE"] and k in table["GRBM_GUI_ACTIVE"] else "-") for k in kernels))
if "TA_TA_BUSY" in table and "GRBM_GUI_ACTIVE" in table:
    print("%-28s" % "TA busy (sum/256/cyc)" + "".join("%*s" % (w, "%.3f" % (table["TA_TA_BUSY"][k] / 256 / (table["GRBM_GUI_ACTIVE"][k] / 8)) if k in table["TA_TA_BUSY"] and k in table["GRBM_GUI_ACTIVE"] else "-") for k in kernels))
