#!/usr/bin/env python3
"""tools/config5_bytes_by_item.py OUTDIR : where config 5's bytes go (review r05 item 6: 85 N measured per limb-product against 56 N
algorithmic and 72 N by the design's own count without L2 retention).  tools/config5_bytes_by_item.sh ran `bench.py --config 5` under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (and the request counters) once per DIAGNOSTIC build in which only ONE item type of
team_product_kernel does its work (-DNTT_TEAMPROD_ONLY=1|8|2|4: b's column items, a's column items, block products, c's inverse
column items; the others only run the queue protocol) and once for the shipped library.  Per limb-product and in units of N bytes."""
import collections, csv, glob, os, sys
out = sys.argv[1]
N, BATCH, LIMBS, STEPS = 1 << 17, 512, 4, 4
UNITS = BATCH * LIMBS * STEPS                      # limb-products per run
NAMES = {"1": "b column items (read b, write col(b))", "8": "a column items (read a, write col(a))",
         "2": "block products (read col(a), col(b) blocks; write c' blocks)", "4": "c inverse column items (read c', write c)",
         "15": "shipped library (all item types)"}
DESIGN = {"1": (8, 8), "8": (8, 8), "2": (16, 8), "4": (8, 8), "15": (40, 32)}


def total(d, ctr):
    s = 0.0
    for f in glob.glob("%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "team_product_kernel" in r["Kernel_Name"]:
                s += float(r["Counter_Value"])
    return s


print("# bytes per limb-product in units of N (N = 2^17 coefficients; 8 N = one polynomial), %d limb-products per run" % UNITS)
print("%-62s %12s %12s %12s   %s" % ("item type doing its work", "FETCH x2 /N", "WRITE /N", "sum /N", "design count (read, write) /N"))
rows = {}
for v in ("1", "8", "2", "4", "15"):
    d = os.path.join(out, "only" + v)
    if not os.path.isdir(d):
        continue
    f = total(d, "FETCH_SIZE") * 1024 * 2 / UNITS / N
    w = total(d, "WRITE_SIZE") * 1024 / UNITS / N
    rows[v] = (f, w)
    print("%-62s %12.2f %12.2f %12.2f   %s" % (NAMES[v], f, w, f + w, DESIGN[v]))
if all(k in rows for k in ("1", "8", "2", "4")):
    f = sum(rows[k][0] for k in ("1", "8", "2", "4"))
    w = sum(rows[k][1] for k in ("1", "8", "2", "4"))
    print("%-62s %12.2f %12.2f %12.2f" % ("sum of the four item types", f, w, f + w))
# request sizes of the shipped library: 32-byte against 64-byte read requests at the L2's memory side
d = os.path.join(out, "only15")
for ctrs in (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")):
    vals = [total(d, c) for c in ctrs]
    if vals[0] > 0:
        print("# shipped library: %s = %.4g, %s = %.4g per run (%.1f %%)" % (ctrs[0], vals[0], ctrs[1], vals[1], 100.0 * vals[1] / vals[0]))
