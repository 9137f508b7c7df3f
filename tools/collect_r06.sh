#!/bin/bash
# tools/collect_r06.sh OUTDIR : what profiles/r06 holds about the SHIPPED library, collected on the GPU box in one go and LAST (review r05
# item 3): every artefact is stamped with the sha256 of libntt_mi355x.so it was taken on (LIBRARY_SHA256, the `lib_sha256` field of every
# bench line and pmc_traffic*.json, a first comment line in the kernel-stats CSVs); tests/test_abi.py compares it with the built library.
#   the default bench line; per BASELINE config the bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE
#   passes (one counter per run) -> per-kernel table and the pmc_traffic*.json files bench.py quotes; the one-pass 2^15 kernel likewise;
#   NTT-domain product kernels; sustained headline with rocm-smi; sweeps; RNS pipeline rows; folded 8-shard rehearsal.
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SHA=$(sha256sum optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so | cut -d' ' -f1)
echo "$SHA  optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so   ($(date -u +%Y-%m-%dT%H:%MZ), $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version.*' | head -1))" > $out/LIBRARY_SHA256
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
for c in 4 2 3 5 5_bm; do
  cfg=${c%_bm}; lay=""; [ $c = 5_bm ] && lay="--layout batch-major"
  # warm-ups cover >= 60 ms of device work (the clocks settle for ~20 ms after the idle gap of the parity check: profiles/r05/clock_settling_after_idle.txt)
  st="--steps 20 --warmup 9"; [ $cfg = 5 ] && st="--steps 10 --warmup 18"; [ $cfg = 2 ] && st="--steps 100 --warmup 80"
  timeout 900 python3 bench.py --config $cfg $lay $st --no-also > $out/bench_config$c.json 2> $out/bench_config$c.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$c -- python3 bench.py --config $cfg $lay $st --no-cpu-baseline --headline-only > $out/bench_config${c}_under_rocprofv3.json 2> $out/kt$c.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc$c/$ctr -- python3 bench.py --config $cfg $lay --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/pmc${c}_$ctr.log 2>&1
  done
done
# config 2 cold: the shape the driver asks the headline for (20 steps behind 3 warm-ups), for the record beside the steady-state figure
timeout 600 python3 bench.py --config 2 --steps 20 --warmup 3 --no-also --no-cpu-baseline > $out/bench_config2_cold_20_steps_3_warmups.json 2>/dev/null
python3 tools/pmc_kernels.py $out > $out/pmc_per_kernel.txt 2>&1
mkdir -p $out/json; python3 tools/pmc_traffic_json.py $out $out/json > $out/pmc_traffic_json.log 2>&1
# the one-pass 2^15 kernel (round 6): kernel trace + traffic counters over a forward + inverse sweep at 2^15 (16 GB slab: 61035 polynomials)
oargs="--logn 15 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --steps 4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_onepass -- python3 tools/sweep.py $oargs > $out/onepass_under_rocprofv3.txt 2> $out/kt_onepass.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_onepass/$ctr -- python3 tools/sweep.py $oargs > $out/pmc_onepass_$ctr.log 2>&1
done
python3 - $out <<'PY' > $out/pmc_onepass_per_kernel.txt 2>&1
import csv, glob, collections, sys
out = sys.argv[1]
N, B = 1 << 15, int(16e9 / 8 / (1 << 15))
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/pmc_onepass/%s/**/*counter_collection.csv" % (out, ctr), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "onepass_kernel" in r["Kernel_Name"]:
                tot["inverse" if ", true," in r["Kernel_Name"].split("onepass_kernel")[1][:40] else "forward"][ctr].append(float(r["Counter_Value"]))
print("# onepass_kernel, N = 2^15, %d polynomials per dispatch: FETCH_SIZE x2 + WRITE_SIZE per transform in units of N bytes (algorithmic: 16 N)" % B)
for d, t in tot.items():
    f = sum(t["FETCH_SIZE"]) / max(len(t["FETCH_SIZE"]), 1) * 1024 * 2 / B / N
    w = sum(t["WRITE_SIZE"]) / max(len(t["WRITE_SIZE"]), 1) * 1024 / B / N
    print("%-8s dispatches %d  FETCH x2 %.3f N  WRITE %.3f N  sum %.3f N  (x %.4f of 16 N)" % (d, len(t["FETCH_SIZE"]), f, w, f + w, (f + w) / 16))
PY
# the NTT-domain product kernels: kernel trace + traffic counters over tools/domain_bench.py
dargs="--logn 14 16 --k 1 3 --steps 4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_dot -- python3 tools/domain_bench.py $dargs > $out/domain_bench_under_rocprofv3.txt 2> $out/kt_dot.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_dot/$ctr -- python3 tools/domain_bench.py $dargs > $out/pmc_dot_$ctr.log 2>&1
done
python3 tools/pmc_dot.py $out > $out/pmc_dot_per_kernel.txt 2>&1
timeout 900 python3 tools/domain_bench.py --logn 8 10 12 13 14 15 16 17 --k 1 2 3 8 > $out/domain_bench.txt 2>&1
# sustained headline: 400 back-to-back steps (about 3 s), rocm-smi power / clocks beside it
bash tools/exp_smi.sh $out/sustained python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --headline-only
mv $out/sustained.log $out/bench_config4_sustained.json; mv $out/sustained.smi $out/rocm_smi_power_clock_during_sustained_bench.txt
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 > $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 12 14 15 16 --ops fwd inv mul --arith auto u64 --qs 0xffffffff00001 --bytes 8e9 > $out/sweep_52bit_modulus.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 17 --ops mul --qs 0x80000001c0001 --bytes 8e9 > $out/sweep_products.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 10 12 13 14 15 16 17 --ops fwd inv mul --qs 0x1fffffffffc0001 0xffffffffffc0001 --bytes 4e9 > $out/sweep_integer_moduli.txt 2>&1
(for lm in "" "--batch-major"; do timeout 300 python3 tools/pipeline_bench.py $lm; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 15 --batch 2048 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096 $lm; done) > $out/pipeline_rns.txt 2>&1
timeout 600 python3 tools/pointer_batch_bench.py > $out/pointer_batches.txt 2>&1
timeout 600 python3 tools/pointer_product_bench.py > $out/pointer_products.txt 2>&1
timeout 300 python3 tools/rns_pointer_small_batch.py > $out/rns_pointer_small_batch.txt 2>&1
timeout 1500 bash tools/folded_8_shards.sh $out/folded_8_shards.txt > /dev/null 2>&1
if [ -x oracle/_ref/ntt-variants-bench-dropin ]; then timeout 600 oracle/_ref/ntt-variants-bench-dropin > $out/reference_bench_driver_dropin.txt 2>&1; fi
timeout 900 python3 tools/soak.py --seconds 300 > $out/soak.txt 2>&1
# kernel-stats CSVs: the sha256 of the library as a first comment line
for c in 4 2 3 5 5_bm onepass; do
  f=$(ls $out/kt$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $out/kt_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && continue
  n=rocprofv3_kernel_stats_config$c.csv; [ $c = 5_bm ] && n=rocprofv3_kernel_stats_config5_batch_major.csv; [ $c = onepass ] && n=rocprofv3_kernel_stats_onepass_2p15.csv
  (echo "# libntt_mi355x.so sha256 $SHA"; cat $f) > $out/$n
done
rm -rf $out/kt*/*/*agent_info.csv
tail -1 $out/bench_default.json | cut -c1-300; cat $out/pmc_per_kernel.txt $out/pmc_onepass_per_kernel.txt $out/pmc_traffic_json.log
