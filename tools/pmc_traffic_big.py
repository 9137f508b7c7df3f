#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per transform for the kernels of the N > 2^14 paths (rocprofv3 --pmc CSVs under ROOT/tpX/{fetch,write})."""
import csv, glob, sys, collections
root = sys.argv[1]
for tp in ("tp0", "tp1"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, grid) -> counter -> sum, plus count
    for ctr in ("fetch", "write"):
        for f in glob.glob("%s/%s/%s/**/*counter_collection.csv" % (root, tp, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "column_kernel" not in k and "fused_kernel" not in k and "twophase_kernel" not in k: continue
                key = (k.split("(")[0][:70], r["Grid_Size"])
                per[key][r["Counter_Name"]] += float(r["Counter_Value"])
                per[key]["n_" + r["Counter_Name"]] += 1
    print("# %s : %s" % (tp, "one launch per pass" if tp == "tp0" else "two-phase kernel (both passes in one workgroup)"))
    for (k, grid), c in sorted(per.items()):
        fs = c["FETCH_SIZE"] / max(c["n_FETCH_SIZE"], 1); ws = c["WRITE_SIZE"] / max(c["n_WRITE_SIZE"], 1)
        print("  %-72s grid %-9s FETCH_SIZE %12.0f KB (x2 gfx950 = %8.1f MiB)  WRITE_SIZE %12.0f KB (%8.1f MiB) per dispatch" % (k, grid, fs, 2 * fs / 1024, ws, ws / 1024))
