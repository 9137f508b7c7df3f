#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
timeout 600 python tools/soak.py --seconds 420 --seed 7 > $out/soak_seed7.txt 2>&1; echo "soak rc=$?" > $out/summary9.txt; tail -3 $out/soak_seed7.txt >> $out/summary9.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1; echo "gpu tests rc=$?" >> $out/summary9.txt; tail -3 $out/tests_gpu.log >> $out/summary9.txt
cat $out/summary9.txt
