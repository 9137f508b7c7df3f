#!/bin/bash
# tools/exp_pmc_team.sh OUTDIR : SQ / TA counters of the one-launch kernels at N = 2^16 side by side: the plain inverse transform
# (team_kernel, forced), the NTT-domain product (team_dot_kernel, k = 1 and 3) and the forward-side product (team_mul_kernel)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_dom/g$i -- python3 tools/domain_bench.py --logn 16 --k 1 3 --steps 3 --no-broadcast --xcd-local 1 > $out/pmc_dom_g$i.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_inv/g$i -- python3 tools/sweep.py --logn 16 --ops inv fwd --bytes 4e9 --steps 3 --xcd-local 1 > $out/pmc_inv_g$i.log 2>&1
done
(echo "== NTT-domain products (tools/domain_bench.py --logn 16 --k 1 3 --xcd-local 1: team_dot_kernel over k = 1 and 3 mixed, team_mul_kernel over its four forms)"; python3 tools/pmc_by_kernel.py $out/pmc_dom team_ 0
 echo "== plain transforms, one launch forced (tools/sweep.py --logn 16 --ops inv fwd --xcd-local 1)"; python3 tools/pmc_by_kernel.py $out/pmc_inv team_ 0) > $out/pmc_team_kernels_2p16.txt 2>&1
rm -rf $out/pmc_*/*/*/*agent_info.csv
cat $out/pmc_team_kernels_2p16.txt
