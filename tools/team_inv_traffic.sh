#!/bin/bash
# tools/team_inv_traffic.sh OUTDIR : FETCH_SIZE / WRITE_SIZE of the INVERSE transform's launches (single-launch form and per-pass
# form) per dispatch, relative to the data of the dispatch
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lg in 16 15; do for x in 1 0; do for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$out/m${lg}_x${x}_$ctr
  timeout 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- python3 tools/sweep.py --logn $lg --ops inv fwd --qs 0x7fffffffe0001 --bytes 4e9 --steps 3 --xcd-local $x --lag 16 > $d.log 2>&1
done; done; done
python3 - $out <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(root + "/m*_x*_*_SIZE")):
    ctr = "FETCH_SIZE" if d.endswith("FETCH_SIZE") else "WRITE_SIZE"
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                per[r["Kernel_Name"].split("(")[0].replace("void ntt::", "")[:70]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        big = [x for x in v if x > 0.3 * max(v)]
        if max(v) > 1e5:
            print("%-28s %-72s %4d dispatches  %s%s = %.1f MiB" % (os.path.basename(d), k, len(big), ctr, " x2" if ctr == "FETCH_SIZE" else "", (2 if ctr == "FETCH_SIZE" else 1) * sum(big) / len(big) / 1024))
PY
