#!/usr/bin/env python3
"""NTT-domain product kernels under rocprofv3 (tools/collect_r05.sh: tools/domain_bench.py --logn 14 16 --k 1 3 --steps 4):
duration from the kernel trace, FETCH_SIZE (x2: gfx950 reports half of a coalesced streaming read, MI355X_MICROARCH.md) and
WRITE_SIZE from the two PMC passes, PER CALL of ntt_inv_dot_batch.  domain_bench.py issues, per size, 6 fused calls (2 warm-ups
+ 4 timed) for each of (k=1), (k=1, broadcast b^), (k=3), (k=3, broadcast b^) in that order: the dispatches of a kernel are
split into those four groups by dispatch order; a call above 2^14 is several dispatches (one per 256 MiB chunk), summed.
usage: tools/pmc_dot.py OUTDIR"""
import collections, csv, glob, re, sys
root = sys.argv[1]
CALLS, GROUPS = 6, ["k=1", "k=1 bcast", "k=3", "k=3 bcast"]


def short(name):
    name = name.split("(")[0].replace("void ntt::", "").replace("ntt::", "")
    return re.sub(r"\s+", " ", name)[:70]


def rows(pattern, value):
    out = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            v = value(r)
            if v is not None:
                out[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), v))
    return {k: [x for _, x in sorted(v)] for k, v in out.items()}


dur = rows("%s/kt_dot/**/*kernel_trace.csv" % root, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
ctr = {c: rows("%s/pmc_dot/%s/**/*counter_collection.csv" % (root, c), lambda r, c=c: float(r["Counter_Value"]) if r["Counter_Name"] == c else None)
       for c in ("FETCH_SIZE", "WRITE_SIZE")}
print("# kernel | operands | dispatches per call | ms per call | FETCH_SIZE x2 MiB per call | WRITE_SIZE MiB per call")
for k in sorted(dur):
    if "dot_inv_kernel" not in k and "team_dot_kernel" not in k:      # (round 5: above 2^14 the whole call is ONE team_dot_kernel launch)
        continue
    d = dur[k]
    if len(d) % (CALLS * len(GROUPS)):
        print("  %s: %d dispatches do not split into %d groups of %d calls" % (k, len(d), len(GROUPS), CALLS))
        continue
    per = len(d) // (CALLS * len(GROUPS))          # dispatches of one call
    for g, name in enumerate(GROUPS):
        lo, hi = g * CALLS * per, (g + 1) * CALLS * per
        ms = sum(d[lo + 2 * per:hi]) / (CALLS - 2)                               # the four timed calls
        line = "  %-70s %-10s %3d  %8.3f" % (k, name, per, ms)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = ctr[c].get(k, [])
            if len(v) == len(d):
                line += "  %10.1f" % ((2 if c == "FETCH_SIZE" else 1) * sum(v[lo:hi]) / CALLS / 1024)
            else:
                line += "  %10s" % "-"
        print(line)
# c^ = fwd(a) (.) b^ (+ c^): domain_bench.py issues 6 fused calls each of (multiply), (multiply, broadcast key), (accumulate), (accumulate, key)
MGROUPS = ["mul", "mul key", "mac", "mac key"]
for k in sorted(dur):
    if "fwd_mul_kernel" not in k and "team_mul_kernel" not in k:
        continue
    d = dur[k]
    if len(d) % (CALLS * len(MGROUPS)):
        print("  %s: %d dispatches do not split into %d groups of %d calls" % (k, len(d), len(MGROUPS), CALLS))
        continue
    per = len(d) // (CALLS * len(MGROUPS))
    for g, name in enumerate(MGROUPS):
        lo, hi = g * CALLS * per, (g + 1) * CALLS * per
        ms = sum(d[lo + 2 * per:hi]) / (CALLS - 2)
        line = "  %-70s %-10s %3d  %8.3f" % (k, name, per, ms)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = ctr[c].get(k, [])
            line += ("  %10.1f" % ((2 if c == "FETCH_SIZE" else 1) * sum(v[lo:hi]) / CALLS / 1024)) if len(v) == len(d) else ("  %10s" % "-")
        print(line)
for k in sorted(dur):
    if "column_kernel" in k or "pointwise_acc" in k:
        d = dur[k]
        print("  %-70s %-10s %3s  %8.3f (avg per dispatch, %d dispatches)" % (k, "", "", sum(d) / len(d), len(d)))
