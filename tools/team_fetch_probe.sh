#!/bin/bash
# tools/team_fetch_probe.sh OUTDIR : does the XCD's L2 keep the intermediate?  FETCH_SIZE of team_kernel per (workgroups per CU, lag)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lg in 16 17; do for wpc in 2 3; do for lag in 2 3 4 5 6 8; do
  d=$out/m${lg}_wpc${wpc}_lag${lag}
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/sweep.py --logn $lg --ops fwd --qs 0x80000001c0001 --bytes 4e9 --steps 3 --xcd-local 1 --lag $lag --wpc $wpc > $d.log 2>&1
done; done; done
python3 tools/team_fetch_summary.py $out
