#!/bin/bash
# tools/exp_pmc_oversub.sh OUTDIR : the 2^12 block kernels with 1 and with 8 workgroups per resident slot under the SQ / TA counters
# (what changes when the four workgroups of a CU no longer run in phase)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q=0x7fffffffe0001
for ov in 1 8; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY GRBM_GUI_ACTIVE" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_ov$ov/g$i -- python3 tools/sweep.py --qs $Q --bytes 8e9 --steps 3 --logn 12 --ops fwd inv --oversub $ov > $out/pmc_ov${ov}_g$i.log 2>&1
  done
  echo "== $ov workgroup(s) per resident slot"; python3 tools/pmc_by_kernel.py $out/pmc_ov$ov fused_kernel 1e9
  rm -rf $out/pmc_ov$ov/*/*/*agent_info.csv
done > $out/pmc_oversub_1_vs_8.txt 2>&1
cat $out/pmc_oversub_1_vs_8.txt
