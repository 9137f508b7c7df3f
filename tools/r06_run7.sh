#!/bin/bash
# scratch: pointer-table forms of the fused product kernels -- parity tests and the bench (fused, and the old chain for comparison)
out=gpurun_out/r06d; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ptr_tables.py -x -q > $out/tests_ptr.log 2>&1; echo "ptr tests rc=$?" > $out/summary.txt
tail -5 $out/tests_ptr.log >> $out/summary.txt
timeout 600 python tools/pointer_product_bench.py > $out/pointer_products.txt 2>&1; echo "bench rc=$?" >> $out/summary.txt
timeout 600 python tools/pointer_product_bench.py --chain > $out/pointer_products_chain.txt 2>&1; echo "chain rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/pointer_products.txt; cat $out/pointer_products_chain.txt
