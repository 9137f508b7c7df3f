#!/bin/bash
# tools/pmc_dot_groups.sh OUTDIR : SQ / TA / VALU counter groups (one rocprofv3 --pmc pass each) over the NTT-domain product kernel
# (k = 1 and k = 8, 2^14) and the forward-times-b^ kernel; summaries by tools/pmc_summary.py
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
declare -A grp=( [sq]="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES" [ta]="TA_TA_BUSY GRBM_GUI_ACTIVE" [valu]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" )
for cfg in "dot1|--op dot --k 1" "dot8|--op dot --k 8" "dot8b|--op dot --k 8 --bcast" "mul|--op mul" "macb|--op mul --acc --bcast"; do
  name="${cfg%%|*}"; args="${cfg#*|}"
  for g in sq ta valu; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/$name/$g -- python3 tools/dot_probe.py $args --launches 4 > $out/${name}_$g.log 2>&1
  done
  echo "## $name: tools/dot_probe.py $args" >> $out/pmc_summary_domain_kernels.txt
  python3 tools/pmc_summary.py $out/$name "$( [ ${name:0:3} = dot ] && echo dot_inv_kernel || echo fwd_mul_kernel )" >> $out/pmc_summary_domain_kernels.txt 2>&1
done
cat $out/pmc_summary_domain_kernels.txt
