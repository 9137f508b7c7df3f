#!/usr/bin/env python3
"""Per kernel of each bench config: average duration from the kernel trace, FETCH_SIZE (x2: gfx950 reports half of a coalesced
streaming read, MI355X_MICROARCH.md) and WRITE_SIZE per dispatch from the two PMC passes.
usage: tools/pmc_kernels.py OUTDIR   (OUTDIR/kt<cfg>, OUTDIR/pmc<cfg>/{FETCH_SIZE,WRITE_SIZE} as tools/collect_r03.sh writes them)"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]


def short(name):
    name = name.split("(")[0].replace("void ntt::", "").replace("ntt::", "")
    return re.sub(r"\s+", " ", name)[:96]


for cfg in ("4", "2", "3", "5", "5_bm"):
    if not glob.glob("%s/kt%s" % (root, cfg)): continue
    dur = collections.defaultdict(list)
    for f in glob.glob("%s/kt%s/**/*kernel_trace.csv" % (root, cfg), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/pmc%s/%s/**/*counter_collection.csv" % (root, cfg, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    ctr[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
    print("# config %s : kernel | launches | avg ms (min) | FETCH_SIZE x2 MiB | WRITE_SIZE MiB  (per dispatch, full-size dispatches)" % cfg)
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d = dur[k]
        big = [x for x in d if x > 0.3 * max(d)]
        line = "  %-96s %4d  %8.3f (%7.3f)" % (k, len(big), sum(big) / len(big), min(big))
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = ctr[k][c]
            if v:
                vb = [x for x in v if x > 0.3 * max(v)]
                line += "  %10.1f" % ((2 if c == "FETCH_SIZE" else 1) * sum(vb) / len(vb) / 1024)
            else:
                line += "  %10s" % "-"
        if sum(d) > 0.02 * sum(sum(v) for v in dur.values()):
            print(line)
