#!/bin/bash
# tools/rebench_after_install.sh OUTDIR [SOAK_SECONDS] : the bench LINES of the collection taken again once tools/install_collection.sh
# has put the collection's pmc_traffic*.json under profiles/r06 -- a line of the collection itself is printed before the counter
# summaries of its own library exist, so its `roofline.traffic` quotes the previous collection's (`traffic_from_this_binary: false`);
# these lines quote the counters of the binary they ran on.  Same commands and step counts as tools/collect_r06.sh.  Then a soak with a
# fresh seed on the same library.
out=$1; soak=${2:-0}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
sha256sum optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so | cut -d' ' -f1 > $out/lib_sha256_of_this_run
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
for c in 4 2 3 5 5_bm; do
  cfg=${c%_bm}; lay=""; [ $c = 5_bm ] && lay="--layout batch-major"
  st="--steps 20 --warmup 9"; [ $cfg = 5 ] && st="--steps 10 --warmup 18"; [ $cfg = 2 ] && st="--steps 100 --warmup 80"
  timeout 900 python3 bench.py --config $cfg $lay $st --no-also > $out/bench_config$c.json 2> $out/bench_config$c.err
done
timeout 600 python3 bench.py --config 2 --steps 20 --warmup 3 --no-also --no-cpu-baseline > $out/bench_config2_cold_20_steps_3_warmups.json 2>/dev/null
grep -o '"traffic_from_this_binary": [a-z]*' $out/bench_*.json
if [ "$soak" -gt 0 ]; then
  timeout $((soak + 300)) python3 tools/soak.py --seconds $soak --seed 13 > $out/soak_seed13.txt 2>&1; echo "soak rc=$?"; tail -3 $out/soak_seed13.txt | cut -c1-250
fi
