#!/bin/bash
# scratch: pointer-table forms of the fused product kernels -- parity tests and the bench
out=gpurun_out/r06d; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ptr_tables.py -x -q > $out/tests_ptr.log 2>&1; echo "ptr tests rc=$?" > $out/summary.txt
tail -5 $out/tests_ptr.log >> $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_parity.py -x -q -k "onepass or one_pass or dot or mul or product or example" > $out/tests_prod.log 2>&1; echo "product tests rc=$?" >> $out/summary.txt
tail -3 $out/tests_prod.log >> $out/summary.txt
timeout 600 python tools/pointer_product_bench.py > $out/pointer_products.txt 2>&1; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/pointer_products.txt
