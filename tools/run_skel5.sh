#!/bin/bash
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 build/skel5 16 12 > $out/skeleton5.txt 2>&1
for sel in "m16 mode0 wpc2 pol1 la1 17 sa1  0 la2  2 sa2 16 F40 TW0" "m17 mode0 wpc2 pol1 la1 17 sa1  0 la2  2 sa2 16 F40 TW0"; do
  tag=$(echo "$sel" | tr -s ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$tag/$c -- build/skel5 4 4 "$sel" > $out/pmc_${tag}_$c.log 2>&1
  done
done
python3 tools/pmc_summary.py $out/pmc k_five > $out/pmc_skel5.txt 2>&1
