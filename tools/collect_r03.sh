#!/bin/bash
# tools/collect_r03.sh OUTDIR : what profiles/r03 holds, collected on the GPU box in one go:
#   per BASELINE config (2, 3, 4, 5): the bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE and
#   WRITE_SIZE passes (one counter per run); size / modulus sweeps; RNS pipeline rows (large and small batches).
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 4 2 3 5; do
  st="--steps 20 --warmup 3"; [ $c = 5 ] && st="--steps 10 --warmup 2"
  timeout 900 python3 bench.py --config $c $st > $out/bench_config$c.json 2> $out/bench_config$c.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$c -- python3 bench.py --config $c $st --no-cpu-baseline --headline-only > $out/bench_config${c}_under_rocprofv3.json 2> $out/kt$c.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc$c/$ctr -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc${c}_$ctr.log 2>&1
  done
done
declare -A grp=( [sq]="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" [ta]="TA_TA_BUSY GRBM_GUI_ACTIVE" [valu]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" )
for g in sq ta valu; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/pmc4/$g -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc4_$g.log 2>&1
done
python3 tools/pmc_kernels.py $out > $out/pmc_per_kernel.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 > $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 15 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --xcd-local 0 > $out/sweep_per_pass.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 15 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --xcd-local 1 > $out/sweep_xcd_local.txt 2>&1
timeout 900 python3 tools/sweep.py --logn 14 --ops fwd inv fwdlazy mul --arith f64 u64 r4 --qs 0x7fffffffe0001 0x80000001c0001 0x3ffffffdf0001 0x7ffe0001 0xffffffff00001 --bytes 4e9 > $out/sweep_arith_moduli.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 12 14 16 --ops fwd inv mul --arith auto u64 --qs 0xffffffff00001 --bytes 8e9 > $out/sweep_52bit_modulus.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops mul --qs 0x80000001c0001 --bytes 8e9 > $out/sweep_products.txt 2>&1
(timeout 300 python3 tools/pipeline_bench.py; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096) > $out/pipeline_rns.txt 2>&1
(for lg in 14 16; do for limbs in 4 16; do for b in 1 2 8 64; do
  for loop in 1 0; do NTT_RNS_LOOP=$loop timeout 120 python3 tools/pipeline_bench.py --logn $lg --limbs $limbs --batch $b --steps 10; done
done; done; done) > $out/pipeline_rns_small_batch.txt 2>&1
if [ -x oracle/_ref/ntt-variants-bench-dropin ]; then timeout 600 oracle/_ref/ntt-variants-bench-dropin > $out/reference_bench_driver_dropin.txt 2>&1; fi
rm -rf $out/kt?/*/*agent_info.csv
tail -1 $out/bench_config4.json | cut -c1-400; cat $out/pmc_per_kernel.txt
