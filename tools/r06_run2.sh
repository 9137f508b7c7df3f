#!/bin/bash
# second GPU call of round 6: pointer tables (tests + measurement), the 2^13 kernel as two independent workgroups per CU (oversub sweep), memset repro v2, strong line
out=gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ptr_tables.py tests/test_gpu_layouts.py -x -q 2>&1 | tail -15
timeout 600 python3 tools/pointer_batch_bench.py > $out/pointer_batches.txt 2>&1; cat $out/pointer_batches.txt
(for ov in 1 2 4 8 16; do echo "== oversub $ov"; timeout 300 python3 tools/sweep.py --logn 13 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --oversub $ov | tail -2; done
 echo "== sizes 11..14, defaults"; timeout 300 python3 tools/sweep.py --logn 11 12 13 14 --ops fwd inv --qs 0x80000001c0001 --bytes 16e9) > $out/sweep_2p13_two_workgroups.txt 2>&1
cat $out/sweep_2p13_two_workgroups.txt
timeout 300 build/memset_graph_repro 12 > $out/memset_graph_repro.txt 2>&1; grep -c "" $out/memset_graph_repro.txt; grep "torch-like\|total" $out/memset_graph_repro.txt
timeout 600 python3 bench.py --gpus 1 --scaling strong --steps 10 --warmup 4 --no-also --cpu-budget-s 3 > $out/bench_config4_strong_one_gpu_128GiB.json 2> $out/strong.err || tail -3 $out/strong.err
cut -c1-400 $out/bench_config4_strong_one_gpu_128GiB.json
