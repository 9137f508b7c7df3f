#!/bin/bash
# tools/collect_artifacts.sh OUTDIR : everything profiles/rNN holds, collected on the GPU box in one go
# (kernel trace + stats, PMC passes one group per run, the plain bench line, the size and modulus sweeps).
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > $out/bench_under_rocprofv3.json 2> $out/kt.log
declare -A grp=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [sq]="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" [ta]="TA_TA_BUSY GRBM_GUI_ACTIVE" [valu]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" )
for g in fetch write sq ta valu; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/pmc/$g/run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc_$g.log 2>&1
done
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1
# HBM-side traffic of the transforms larger than one block (N = 2^16, 2^17), one launch per pass and two-phase
for tp in 0 1; do for g in fetch write; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/pmc_big/tp$tp/$g/run -- python3 tools/sweep.py --logn 16 17 --ops fwd --qs 0x7fffffffe0001 --bytes 4e9 --steps 2 --two-phase $tp > $out/pmc_big_tp${tp}_$g.log 2>&1
done; done
python3 tools/pmc_traffic_big.py $out/pmc_big > $out/pmc_traffic_two_pass.txt 2>&1
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout 900 python3 bench.py --scaling strong --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_strong_1gpu.json 2>> $out/bench.err
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 > $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 | tail -2 >> $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --two-phase 1 > $out/sweep_two_phase.txt 2>&1
timeout 900 python3 tools/sweep.py --logn 14 --ops fwd inv fwdlazy mul --arith f64 u64 r4 --qs 0x7fffffffe0001 0x80000001c0001 0x3ffffffdf0001 0x7ffe0001 0xffffffff00001 --bytes 4e9 > $out/sweep_arith_moduli.txt 2>&1
(timeout 300 python3 tools/pipeline_bench.py; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096) > $out/pipeline_rns.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 12 14 16 --ops fwd inv mul --arith auto u64 --qs 0xffffffff00001 --bytes 8e9 > $out/sweep_52bit_modulus.txt 2>&1
timeout 600 build/skel 16 16 > $out/skeleton.txt 2>&1
if [ -x oracle/_ref/ntt-variants-bench-dropin ]; then timeout 600 oracle/_ref/ntt-variants-bench-dropin > $out/reference_bench_driver_dropin.txt 2>&1; fi
tail -1 $out/bench.json; cat $out/pmc_summary.txt $out/pmc_traffic_two_pass.txt
