#!/bin/bash
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SKEL4_SWEEP2=1
timeout 900 build/skel4 16 12 > $out/skeleton4_sweep2.txt 2>&1
for sel in "m16 mode0 wpc2 lag6 la1  2 sa1  0 la2  2 sa2  0 F0  X0 TW0" "m16 mode0 wpc2 lag6 la1  2 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m16 mode0 wpc2 lag4 la1  2 sa1  0 la2  2 sa2 18 F0  X0 TW0" "m16 mode0 wpc1 lag3 la1  2 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m16 mode0 wpc1 lag3 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m16 mode0 wpc2 lag4 la1  3 sa1  0 la2  2 sa2  2 F0  X0 TW0"; do
  tag=$(echo "$sel" | tr -s ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$tag/$c -- build/skel4 4 4 "$sel" > $out/pmc_${tag}_$c.log 2>&1
  done
done
python3 tools/pmc_summary.py $out/pmc k_four > $out/pmc_skel4_sweep2.txt 2>&1
cat $out/skeleton4_sweep2.txt; cat $out/pmc_skel4_sweep2.txt
