#!/bin/bash
# tools/collect_r05.sh OUTDIR : what profiles/r05 holds, collected on the GPU box in one go:
#   the default bench line (headline + also_config2/3/5 [+ config 5 in [batch][limb][N] layout]); per BASELINE config the bench line,
#   rocprofv3 --kernel-trace --stats of the same command and FETCH_SIZE / WRITE_SIZE passes (one counter per run) -> per-kernel table and
#   the pmc_traffic*.json files bench.py quotes; the same for the NTT-domain product kernels; the sustained headline with rocm-smi
#   beside it; sweeps; RNS pipeline rows in both layouts.
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
for c in 4 2 3 5 5_bm; do
  cfg=${c%_bm}; lay=""; [ $c = 5_bm ] && lay="--layout batch-major"
  # warm-ups cover >= 60 ms of device work (the clocks settle for ~20 ms after the idle gap of the parity check: clock_settling_after_idle.txt)
  st="--steps 20 --warmup 9"; [ $cfg = 5 ] && st="--steps 10 --warmup 18"; [ $cfg = 2 ] && st="--steps 100 --warmup 80"
  timeout 900 python3 bench.py --config $cfg $lay $st --no-also > $out/bench_config$c.json 2> $out/bench_config$c.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$c -- python3 bench.py --config $cfg $lay $st --no-cpu-baseline --headline-only > $out/bench_config${c}_under_rocprofv3.json 2> $out/kt$c.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc$c/$ctr -- python3 bench.py --config $cfg $lay --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc${c}_$ctr.log 2>&1
  done
done
python3 tools/pmc_kernels.py $out > $out/pmc_per_kernel.txt 2>&1
mkdir -p $out/json; python3 tools/pmc_traffic_json.py $out $out/json > $out/pmc_traffic_json.log 2>&1
# the NTT-domain product kernels: kernel trace + traffic counters over tools/domain_bench.py (2^14: one launch; 2^16: one launch over both passes)
dargs="--logn 14 16 --k 1 3 --steps 4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_dot -- python3 tools/domain_bench.py $dargs > $out/domain_bench_under_rocprofv3.txt 2> $out/kt_dot.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_dot/$ctr -- python3 tools/domain_bench.py $dargs > $out/pmc_dot_$ctr.log 2>&1
done
python3 tools/pmc_dot.py $out > $out/pmc_dot_per_kernel.txt 2>&1
timeout 900 python3 tools/domain_bench.py --logn 8 10 12 13 14 15 16 17 --k 1 2 3 8 > $out/domain_bench.txt 2>&1
timeout 600 python3 tools/domain_bench.py --logn 12 14 16 --k 1 3 --bits 51 52 60 --no-broadcast > $out/domain_bench_moduli.txt 2>&1
# sustained headline: 400 back-to-back steps (about 3 s), rocm-smi power / clocks beside it
bash tools/exp_smi.sh $out/sustained python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --headline-only
mv $out/sustained.log $out/bench_config4_sustained.json; mv $out/sustained.smi $out/rocm_smi_power_clock_during_sustained_bench.txt
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 > $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 12 14 16 --ops fwd inv mul --arith auto u64 --qs 0xffffffff00001 --bytes 8e9 > $out/sweep_52bit_modulus.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops mul --qs 0x80000001c0001 --bytes 8e9 > $out/sweep_products.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 10 12 13 14 15 16 17 --ops fwd inv mul --qs 0x1fffffffffc0001 0xffffffffffc0001 --bytes 4e9 > $out/sweep_integer_moduli.txt 2>&1
(for lm in "" "--batch-major"; do timeout 300 python3 tools/pipeline_bench.py $lm; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 15 --batch 2048 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096 $lm; done) > $out/pipeline_rns.txt 2>&1
(for lg in 14 16; do for limbs in 4 16; do for b in 1 2 8 64; do
  for loop in 1 0; do NTT_RNS_LOOP=$loop timeout 120 python3 tools/pipeline_bench.py --logn $lg --limbs $limbs --batch $b --steps 10; done
done; done; done) > $out/pipeline_rns_small_batch.txt 2>&1
if [ -x oracle/_ref/ntt-variants-bench-dropin ]; then timeout 600 oracle/_ref/ntt-variants-bench-dropin > $out/reference_bench_driver_dropin.txt 2>&1; fi
rm -rf $out/kt*/*/*agent_info.csv
tail -1 $out/bench_default.json | cut -c1-300; cat $out/pmc_per_kernel.txt $out/pmc_dot_per_kernel.txt $out/pmc_traffic_json.log
