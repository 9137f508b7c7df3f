#!/bin/bash
# tools/collect_dot_pmc.sh OUTDIR : the NTT-domain product kernels under rocprofv3 (the part of tools/collect_r05.sh that covers them):
# kernel trace + FETCH_SIZE / WRITE_SIZE passes over tools/domain_bench.py at 2^14 (one launch) and 2^16 (one launch over both passes)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
dargs="--logn 14 16 --k 1 3 --steps 4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_dot -- python3 tools/domain_bench.py $dargs > $out/domain_bench_under_rocprofv3.txt 2> $out/kt_dot.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_dot/$ctr -- python3 tools/domain_bench.py $dargs > $out/pmc_dot_$ctr.log 2>&1
done
python3 tools/pmc_dot.py $out > $out/pmc_dot_per_kernel.txt 2>&1
timeout 900 python3 tools/domain_bench.py --logn 8 10 12 13 14 15 16 17 --k 1 2 3 8 > $out/domain_bench.txt 2>&1
timeout 600 python3 tools/domain_bench.py --logn 12 14 16 --k 1 3 --bits 51 52 60 --no-broadcast > $out/domain_bench_moduli.txt 2>&1
rm -rf $out/kt*/*/*agent_info.csv
cat $out/pmc_dot_per_kernel.txt
