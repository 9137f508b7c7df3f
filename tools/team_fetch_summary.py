#!/usr/bin/env python3
"""FETCH_SIZE (x2, gfx950) of team_kernel per dispatch relative to the data size, and the throughput sweep.py printed beside it"""
import csv, glob, os, re, sys
root = sys.argv[1]
for d in sorted(glob.glob(root + "/m*_wpc*_lag*")):
    if not os.path.isdir(d): continue
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "team_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(r["Counter_Value"]))
    log = open(d + ".log").read() if os.path.exists(d + ".log") else ""
    m = re.search(r"fwd\s+(\d+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)", log)
    if not vals or not m: print(os.path.basename(d), "no data"); continue
    batch = int(m.group(1)); logn = int(re.search(r"m(\d+)_", os.path.basename(d)).group(1))
    data = batch * (8 << logn)
    print("%-22s FETCH x2 / data = %.2f   (under the profiler: %s of the roofline)" % (os.path.basename(d), 2 * 1024 * sum(vals) / len(vals) / data, m.group(4)))
