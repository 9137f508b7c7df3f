#!/bin/bash
# sixth GPU call of round 6: config 5 with L2-hot twiddles (diagnostic), then the evidence collection on the shipped library
out=gpurun_out/r06; mkdir -p $out/config5_items
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export NTT_BENCH_NOCHECK=1 NTT_LIB=build/libntt_fakeblk.so
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/config5_items/fakeblk/$ctr -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/config5_items/fakeblk_$ctr.log 2>&1
done
unset NTT_BENCH_NOCHECK NTT_LIB
python3 - <<'PY' > gpurun_out/r06/config5_fakeblk.txt
import csv, glob
N, UNITS = 1 << 17, 512 * 4 * 4
for ctr, mul in (("FETCH_SIZE", 2), ("WRITE_SIZE", 1)):
    s = 0.0
    for f in glob.glob("gpurun_out/r06/config5_items/fakeblk/%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "team_product_kernel" in r["Kernel_Name"]:
                s += float(r["Counter_Value"])
    print("all item types, every block product reading block 0's twiddles (L2-hot): %s x%d = %.2f N per limb-product" % (ctr, mul, s * 1024 * mul / UNITS / N))
PY
cat $out/config5_fakeblk.txt
rm -rf $out/config5_items/*/*/*/*agent_info.csv $out/config5_items/*/*/*/*kernel_trace.csv 2>/dev/null
timeout 3000 bash tools/collect_r06.sh gpurun_out/r06c 2>&1 | tail -60
