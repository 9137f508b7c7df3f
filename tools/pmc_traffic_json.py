#!/usr/bin/env python3
"""tools/pmc_traffic_json.py OUTDIR DEST [STEPS_PER_RUN]: the committed traffic figures bench.py quotes (roofline.traffic and the
also_configN blocks' `traffic`), from the FETCH_SIZE / WRITE_SIZE passes tools/collect_r06.sh takes per BASELINE config
(OUTDIR/pmc<cfg>[_bm]/{FETCH_SIZE,WRITE_SIZE}: one counter per run, `bench.py --config N --steps 3 --warmup 1 --headline-only`, so
every dispatch of a library kernel belongs to one of STEPS_PER_RUN = 4 identical steps).
  config 4 -> DEST/pmc_traffic.json                 per LAUNCH of the headline kernel (the r01..r04 format)
  config 2, 3, 5 (and 5 in the [batch][limb][N] layout) -> DEST/pmc_traffic_config<N>[_batch_major].json   per STEP, all kernels
FETCH_SIZE is doubled (gfx950 reports half of a coalesced streaming read: MI355X_MICROARCH.md); WRITE_SIZE is exact."""
import collections, csv, glob, hashlib, json, os, re, sys
out, dest = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd", "libntt_mi355x.so")
LIB_SHA = hashlib.sha256(open(LIB, "rb").read()).hexdigest()    # the binary the passes were taken on: bench.py and tests/test_abi.py compare it
steps_per_run = int(sys.argv[3]) if len(sys.argv) > 3 else 4
SHAPES = {2: (1 << 12, 65536), 3: (1 << 16, 8192), 4: (1 << 14, 131072), 5: (1 << 17, 512)}
BYTES = {2: 16 << 12, 3: 32 << 16, 4: 16 << 14, 5: 4 * (56 << 17)}


def short(name):
    m = re.search(r"(\w+)<ntt::(\w+(?:<\d+>)?), (\d+), (true|false)", name)
    if m and m.group(1) == "fused_kernel":
        return "%s<%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), "inv" if m.group(4) == "true" else "fwd")
    return re.sub(r"\s+", " ", name.split("(")[0].replace("void ntt::", "").replace("ntt::", ""))[:80]


def totals(d):
    """counter -> kernel -> [sum KB, dispatches] over the library's transform / product kernels"""
    t = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        per = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] != ctr or "ntt::" not in k or "fill_uniform" in k or "probe" in k or "checksum" in k: continue
                per[short(k)][0] += float(r["Counter_Value"]); per[short(k)][1] += 1
        t[ctr] = per
    return t


for cfg, suffix in ((4, ""), (2, ""), (3, ""), (5, ""), (5, "_bm")):
    d = "%s/pmc%d%s" % (out, cfg, suffix)
    if not os.path.isdir(d): continue
    t = totals(d)
    n, batch = SHAPES[cfg]
    if cfg == 4:
        k = "fused_kernel<ArithF64,14,fwd>"
        fs, fn = t["FETCH_SIZE"][k]; ws, wn = t["WRITE_SIZE"][k]
        fkb, wkb = round(fs / fn), round(ws / wn)
        j = {"source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_r06.sh), mean over the %d full-size dispatches of %s "
                        "(library sha256 " + LIB_SHA[:16] + "), batch %d") % (fn, k, batch),
             "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "gfx950_fetch_correction": 2.0,
             "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of coalesced streaming reads; WRITE_SIZE is exact",
             "hbm_bytes_per_launch": 2 * 1024 * fkb + 1024 * wkb, "batch": batch, "N": n, "kernel": k}
        name = "pmc_traffic.json"
    else:
        fetch = sum(v[0] for v in t["FETCH_SIZE"].values()) / steps_per_run
        write = sum(v[0] for v in t["WRITE_SIZE"].values()) / steps_per_run
        hbm = int(2 * 1024 * fetch + 1024 * write)
        j = {"source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_r06.sh) over `bench.py --config %d%s --steps 3 --warmup 1 "
                        "--headline-only`: sum over every library kernel of the run / %d steps (library sha256 " + LIB_SHA[:16] + ")") % (cfg, " --layout batch-major" if suffix else "", steps_per_run),
             "config": cfg, "layout": "[batch][limb][N]" if suffix else None, "N": n, "batch": batch,
             "FETCH_SIZE_KB_per_step": round(fetch), "WRITE_SIZE_KB_per_step": round(write), "gfx950_fetch_correction": 2.0,
             "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": batch * BYTES[cfg], "ratio": round(hbm / (batch * BYTES[cfg]), 4),
             "kernels": {k: {"dispatches_per_step": v[1] / steps_per_run, "FETCH_SIZE_x2_MiB_per_step": round(2 * v[0] / 1024 / steps_per_run, 1),
                             "WRITE_SIZE_MiB_per_step": round(t["WRITE_SIZE"][k][0] / 1024 / steps_per_run, 1)} for k, v in t["FETCH_SIZE"].items()}}
        name = "pmc_traffic_config%d%s.json" % (cfg, "_batch_major" if suffix else "")
    j["lib_sha256"] = LIB_SHA
    with open(os.path.join(dest, name), "w") as f:
        json.dump(j, f, indent=1)
    print(name, j.get("hbm_bytes_per_launch") or j.get("hbm_bytes_per_step"), j.get("ratio", ""))
