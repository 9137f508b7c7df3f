#!/bin/bash
# tools/exp_blk12.sh OUTDIR : where the 2^12 block kernel's time goes, next to the 2^10 and 2^14 kernels (profiles/r05/blk12_*):
#   memory / VALU / exchange skeletons of both shapes (build/skel12), A/B builds of the 2^12 forward loop (whole-line stores,
#   plain loads), resident-workgroup sweep, per-phase stamps at 2^12 and 2^14, PMC groups of the three forward kernels side by side
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q=0x7fffffffe0001
S="python3 tools/sweep.py --qs $Q --bytes 8e9 --steps 10"
[ -x build/skel12 ] && timeout 300 build/skel12 8 12 > $out/skel12.txt 2>&1
timeout 300 $S --logn 10 12 14 --ops fwd inv > $out/sweep_base.txt 2>&1
for rep in 1 2; do
  for v in "" wl12 nont; do
    lib=optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so; [ -n "$v" ] && lib=build/libntt_$v.so
    [ -f $lib ] || continue
    echo "== rep $rep variant ${v:-shipped}"; NTT_LIB=$lib timeout 300 $S --logn 12 14 --ops fwd | tail -2
  done
done > $out/ab_variants.txt 2>&1
for g in 256 512 768 1024 1536 2048; do echo "== max grid $g"; timeout 300 $S --logn 12 --ops fwd inv --max-grid $g | tail -2; done > $out/grid_sweep_2p12.txt 2>&1
if [ -f build/libntt_stamps.so ]; then for lg in 12 14; do NTT_LIB=build/libntt_stamps.so timeout 300 python3 tools/stamps.py $lg; done > $out/phase_stamps_2p12_2p14.txt 2>&1; fi
rocprofv3 -L > $out/rocprofv3_counter_list.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc/g$i -- python3 tools/sweep.py --qs $Q --bytes 8e9 --steps 3 --logn 10 12 14 --ops fwd > $out/pmc_g$i.log 2>&1
done
python3 tools/pmc_by_kernel.py $out/pmc fused_kernel 1e9 > $out/pmc_blk.txt 2>&1
rm -rf $out/pmc/*/*/*agent_info.csv
cat $out/skel12.txt $out/sweep_base.txt $out/ab_variants.txt $out/grid_sweep_2p12.txt $out/phase_stamps_2p12_2p14.txt $out/pmc_blk.txt
