#!/bin/bash
# third GPU call of round 6: the one-pass 2^15 kernel (tests, A/B), 2^13 shapes A/B, 2^11 grid sweep, memset repro under torch's runtime
out=gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_onepass.py tests/test_gpu_ptr_tables.py -x -q 2>&1 | tail -15
Q=0x80000001c0001
(echo "# N = 2^15: one pass (onepass_kernel) against the two-pass forms, same box, alternating; 16 GB slabs"
 for rep in 1 2; do
  for op in 1 0; do echo "== one_pass $op"; timeout 300 python3 tools/sweep.py --logn 15 --ops fwd inv --qs $Q --bytes 16e9 --one-pass $op | tail -2; done
  echo "== one_pass 1, inverse halves preload their twiddles (20 spilled VGPRs: build/libntt_invpre.so)"; NTT_LIB=build/libntt_invpre.so timeout 300 python3 tools/sweep.py --logn 15 --ops fwd inv --qs $Q --bytes 16e9 --one-pass 1 | tail -2
 done
 echo "== 52-bit modulus"; for op in 1 0; do timeout 300 python3 tools/sweep.py --logn 15 --ops fwd inv --qs 0xffffffff00001 --bytes 16e9 --one-pass $op | tail -2; done
 echo "== small batches (polynomials: bytes/2^18)"; for b in 64e6 128e6 256e6 1e9; do for op in 1 0; do echo "bytes $b one_pass $op"; timeout 300 python3 tools/sweep.py --logn 15 --ops fwd inv --qs $Q --bytes $b --one-pass $op --steps 30 | tail -2; done; done
) > $out/onepass_2p15.txt 2>&1
cat $out/onepass_2p15.txt
(echo "# 2^13 forward: two independent 512-thread workgroups per CU (tree) against one 1024-thread workgroup of two halves (build/libntt_persist2.so), alternating"
 for rep in 1 2 3; do
  echo "== tree"; timeout 300 python3 tools/sweep.py --logn 13 --ops fwd --qs $Q --bytes 16e9 | tail -1
  echo "== persist2"; NTT_LIB=build/libntt_persist2.so timeout 300 python3 tools/sweep.py --logn 13 --ops fwd --qs $Q --bytes 16e9 | tail -1
 done) > $out/ab_2p13_shapes.txt 2>&1
cat $out/ab_2p13_shapes.txt
(echo "# 2^11: workgroups per resident slot (NTT_OPT_BLOCK_OVERSUB on the table-sharing small-block kernels; default 4)"
 for ov in 1 2 4 8 16 32 64; do echo "== per slot $ov"; timeout 300 python3 tools/sweep.py --logn 11 --ops fwd inv --qs $Q --bytes 16e9 --oversub $ov | tail -2; done) > $out/sweep_2p11_grid.txt 2>&1
cat $out/sweep_2p11_grid.txt
TL=$(python3 -c 'import os,torch;print(os.path.join(os.path.dirname(torch.__file__),"lib"))' 2>/dev/null)
(echo "# the same binary under the HIP runtime the torch wheel bundles ($TL): the failing library test of round 5 ran on THAT runtime (torch imported first)"
 LD_LIBRARY_PATH=$TL timeout 300 build/memset_graph_repro 12) > $out/memset_graph_repro_torch_runtime.txt 2>&1
head -3 $out/memset_graph_repro_torch_runtime.txt; grep -c "replays wrong" $out/memset_graph_repro_torch_runtime.txt; grep -v " 0 of 12 replays" $out/memset_graph_repro_torch_runtime.txt | head -20
timeout 600 python3 tools/pointer_batch_bench.py > $out/pointer_batches.txt 2>&1; cat $out/pointer_batches.txt
