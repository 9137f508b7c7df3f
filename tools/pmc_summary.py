#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per counter, the mean over the full-size dispatches of the fused kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "fused_kernel"
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if not rows: continue
    # full-size dispatches = those with the most common largest Grid_Size * are benchmark steps; probes are tiny
    by_disp = collections.defaultdict(dict)
    for r in rows:
        by_disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        by_disp[r["Dispatch_Id"]]["_grid"] = int(r["Grid_Size"]); by_disp[r["Dispatch_Id"]]["_vgpr"] = r.get("VGPR_Count"); by_disp[r["Dispatch_Id"]]["_lds"] = r.get("LDS_Block_Size")
    # keep dispatches whose first counter value is within 2x of the max (drops the 2-polynomial parity probe)
    names = [k for k in next(iter(by_disp.values())) if not k.startswith("_")]
    key = names[0]
    mx = max(d[key] for d in by_disp.values())
    full = [d for d in by_disp.values() if d[key] > 0.5 * mx]
    print("# %s : %d full dispatches, grid %s rocprofv3-VGPR_Count %s (= half the code object's vgpr_count for these wave64 kernels; tools/check_spills.py prints the ELF value) lds %s" % ([x for x in f.split("/") if x not in ("run","runc")][-2], len(full), full[0]["_grid"], full[0]["_vgpr"], full[0]["_lds"]))
    for n in names:
        print("  %-34s %16.0f" % (n, sum(d[n] for d in full) / len(full)))
