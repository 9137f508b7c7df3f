#!/bin/bash
# tools/pmc_team.sh OUTDIR : VALU-busy counters of the XCD-local kernels (configs 3 and 5), same groups as config 4's passes
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 5 3; do
  for g in valu ta; do
    [ $g = valu ] && ctrs="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" || ctrs="TA_TA_BUSY GRBM_GUI_ACTIVE"
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$c/$g -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc${c}_$g.log 2>&1
  done
done
for c in 5 3; do for pat in team_product_kernel team_kernel; do echo "## config $c $pat"; python3 tools/pmc_summary.py $out/pmc$c $pat; done; done
