#!/bin/bash
# tools/run_skeletons.sh OUTDIR : the XCD-local two-pass skeleton sweeps of profiles/r03 (GPU box; make skel4 skel5 first)
#   skel4 sweep 1 (default)  lag / residency, plain stores            -> skel4_sweep1.txt
#   SKEL4_SWEEP2             cache-policy hints on the streaming sides -> skel4_sweep2.txt
#   SKEL4_SWEEP3             lag in single steps, assignment, exchanges -> skel4_sweep3.txt
#   SKEL4_SWEEP4             items shaped like the library's passes     -> skel4_sweep4.txt
#   SKEL4_SWEEP5             the same with the next item's loads in flight -> skel4_sweep5.txt
#   skel5                    barrier-free items, class queues          -> skel5.txt
# and FETCH_SIZE / WRITE_SIZE passes (one counter per run) of selected rows -> pmc/
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 build/skel4 16 16 > $out/skel4_sweep1.txt 2>&1
for n in 2 3 4 5; do env SKEL4_SWEEP$n=1 timeout 900 build/skel4 16 12 > $out/skel4_sweep$n.txt 2>&1; done
timeout 900 build/skel5 16 12 > $out/skel5.txt 2>&1
for sel in "m16 mode0 wpc2 lag6  blk0 la1  2 sa1  0 la2  2 sa2  0 F0  X0 TW0" "m16 mode0 wpc2 lag6  blk0 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0" \
           "m17 mode0 wpc2 lag3  blk0 la1 17 sa1  0 la2  2 sa2 16 F0  X0 TW0"; do
  tag=$(echo "$sel" | tr -s ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$tag/$c -- build/skel4 4 4 "$sel" > $out/pmc_${tag}_$c.log 2>&1
  done
done
python3 tools/pmc_summary.py $out/pmc k_four > $out/pmc_skel4.txt 2>&1
