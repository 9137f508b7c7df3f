#!/bin/bash
# tools/exp_stagger.sh OUTDIR : do the four 2^12 workgroups of a CU stay apart when they START apart?  build/libntt_stag{3,6}.so
# (-DNTT_STAGGER: the persistent loops begin (TG_ID & 3) * 3 or 6 * 1024 clocks late) against the shipped library at 1, 2 and 8
# workgroups per resident slot, 8 GB per launch and config 2's own 2 GiB; tools/hwid_probe first (what TG_ID says)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 60 ./build/hwid_probe 1024 > $out/hwid_probe.txt 2>&1; tail -14 $out/hwid_probe.txt
(for rep in 1 2; do for bytes in 8e9 2.147e9; do for lib in "" build/libntt_stag3.so build/libntt_stag6.so; do for ov in 1 2 8; do
  echo "rep $rep bytes $bytes ${lib:-shipped} oversub $ov: $(NTT_LIB=$lib timeout 300 python3 tools/sweep.py --logn 12 --ops fwd inv --qs 0x7fffffffe0001 0x3ffffffffc001 --bytes $bytes --steps 20 --oversub $ov | tail -n +2 | awk '{printf "%s %s %s | ", $2, $4, $8}')"
done; done; done; done) > $out/stagger.txt 2>&1
cat $out/stagger.txt
