#!/bin/bash
# tools/collect_artifacts.sh OUTDIR : everything profiles/rNN holds, collected on the GPU box in one go
# (kernel trace + stats, PMC passes one group per run, the plain bench line, the size and modulus sweeps).
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.log
declare -A grp=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [sq]="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" [ta]="TA_TA_BUSY GRBM_GUI_ACTIVE" [valu]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" )
for g in fetch write sq ta valu; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${grp[$g]} --output-format csv -d $out/pmc/$g/run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$g.log 2>&1
done
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout 600 python3 tools/sweep.py --logn 8 9 10 11 12 13 14 15 16 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 > $out/sweep_sizes.txt 2>&1
timeout 600 python3 tools/sweep.py --logn 17 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 | tail -2 >> $out/sweep_sizes.txt 2>&1
timeout 900 python3 tools/sweep.py --logn 14 --ops fwd inv mul --arith f64 u64 --qs 0x7fffffffe0001 0x80000001c0001 0x3ffffffdf0001 0x7ffe0001 --bytes 4e9 > $out/sweep_arith.txt 2>&1
(timeout 300 python3 tools/pipeline_bench.py; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096) > $out/pipeline_rns.txt 2>&1
tail -1 $out/bench.json; cat $out/pmc_summary.txt
