#!/bin/bash
# tools/exp_grid.sh OUTDIR : does the number of workgroups of the persistent block kernels matter?  (the 2^12 forward kernel ran
# +2 % with twice the resident count in the first r05 run)  Alternating repetitions on one box; shipped library and the
# whole-line-store build of the 2^12 forward loop.
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q=0x7fffffffe0001
S="python3 tools/sweep.py --qs $Q --bytes 8e9 --steps 10"
for rep in 1 2 3; do
  for g in 1024 1280 1536 2048 3072 4096 8192 16384 65536; do echo "rep $rep 2^12 grid $g: $(timeout 200 $S --logn 12 --ops fwd inv --max-grid $g | tail -2 | awk '{printf "%s %s  ", $4, $8}')"; done
  for g in 256 320 384 512 768 1024 2048 8192; do echo "rep $rep 2^14 grid $g: $(timeout 200 $S --logn 14 --ops fwd inv --max-grid $g | tail -2 | awk '{printf "%s %s  ", $4, $8}')"; done
  if [ -f build/libntt_wl12.so ]; then
    for g in 1024 2048 4096; do echo "rep $rep 2^12 whole-line stores grid $g: $(NTT_LIB=build/libntt_wl12.so timeout 200 $S --logn 12 --ops fwd --max-grid $g | tail -1 | awk '{printf "%s %s  ", $4, $8}')"; done
  fi
  for lg in 10 13; do for g in 0 4096 16384; do echo "rep $rep 2^$lg grid $g: $(timeout 200 $S --logn $lg --ops fwd inv --max-grid $g | tail -2 | awk '{printf "%s %s  ", $4, $8}')"; done; done
done > $out/grid_sweep.txt 2>&1
cat $out/grid_sweep.txt
