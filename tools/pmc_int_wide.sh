#!/bin/bash
# tools/pmc_int_wide.sh OUTDIR : kernel durations (rocprofv3 --kernel-trace --stats) and the SQ / VALU / HBM counter groups (one
# --pmc pass each) of the integer block kernels at 2^14, 57-bit modulus: ArithU64X<3> (NTT_INT_WIDE=1) and the reference's
# butterflies (NTT_INT_WIDE=0); tools/int_policy_probe.py is the workload (4 GiB slab, 3 + 10 launches per direction)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
declare -A grp=( [sq]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" [valu]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" )
for w in 1 0; do
  export NTT_INT_WIDE=$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/w$w/stats -- python3 tools/int_policy_probe.py --arith auto --bits 57 --logn 14 > $out/w${w}_stats.log 2>&1
  for g in sq valu fetch write; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/w$w/$g -- python3 tools/int_policy_probe.py --arith auto --bits 57 --logn 14 --steps 2 > $out/w${w}_$g.log 2>&1
  done
  echo "## NTT_INT_WIDE=$w  (2^14, 57-bit q, 32768 polynomials per launch; FETCH_SIZE / WRITE_SIZE in KiB, FETCH x 2 on gfx950)" >> $out/pmc_summary_int_wide.txt
  grep "^2\^" $out/w${w}_stats.log >> $out/pmc_summary_int_wide.txt
  f=$(ls $out/w$w/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep "fused_kernel" $f | head -4 >> $out/pmc_summary_int_wide.txt
  python3 tools/pmc_summary.py $out/w$w fused_kernel >> $out/pmc_summary_int_wide.txt 2>&1
done
cat $out/pmc_summary_int_wide.txt
