#!/bin/bash
# tools/prof.sh OUTDIR "CTR CTR ..." ["CTR ..."]...  : one rocprofv3 --pmc pass per counter group over bench.py
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc$i.log 2>&1
done
python3 tools/pmc_summary.py $out
