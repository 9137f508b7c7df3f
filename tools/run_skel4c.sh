#!/bin/bash
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SKEL4_SWEEP3=1
timeout 900 build/skel4 16 12 > $out/skeleton4_sweep3.txt 2>&1
for sel in "m17 mode0 wpc2 lag3  blk0 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m17 mode0 wpc2 lag6  blk0 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m16 mode0 wpc2 lag6  blk0 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0" "m16 mode0 wpc2 lag6  blk1 la1 17 sa1  0 la2  2 sa2 16 F40 X1 TW1"; do
  tag=$(echo "$sel" | tr -s ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$tag/$c -- build/skel4 4 4 "$sel" > $out/pmc_${tag}_$c.log 2>&1
  done
done
python3 tools/pmc_summary.py $out/pmc k_four > $out/pmc_skel4_sweep3.txt 2>&1
