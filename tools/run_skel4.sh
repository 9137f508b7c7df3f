#!/bin/bash
# tools/run_skel4.sh OUTDIR : skeleton sweep + fabric-traffic counters of the best-guess fused row and of the two-launch baseline
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 build/skel4 16 16 > $out/skeleton4.txt 2>&1
for sel in "m16 mode0 wpc4 lag3 la1  2 sa1  0 la2  2 sa2  0 F0  X0 TW0" "m16 mode1 wpc4 lag0 la1  2 sa1  0 la2  2 sa2  0 F0  X0 TW0" "m17 mode0 wpc4 lag3 la1  2 sa1  0 la2  2 sa2  0 F0  X0 TW0"; do
  tag=$(echo "$sel" | tr -s ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$tag/$c -- build/skel4 4 4 "$sel" > $out/pmc_${tag}_$c.log 2>&1
  done
done
python3 tools/pmc_summary.py $out/pmc k_four > $out/pmc_skel4.txt 2>&1
cat $out/skeleton4.txt; cat $out/pmc_skel4.txt
