#!/bin/bash
# scratch: GPU suite, soak, then the evidence collection on the shipped library
mkdir -p gpurun_out/r06i
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06i_tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r06i_tests_gpu.log
timeout 700 python tools/soak.py --seconds 240 --seed 10 > gpurun_out/r06i_soak_seed10.txt 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r06i_soak_seed10.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_r06.sh gpurun_out/r06i > gpurun_out/collect_r06i.log 2>&1
tail -3 gpurun_out/collect_r06i.log; cat gpurun_out/r06i/LIBRARY_SHA256; head -c 400 gpurun_out/r06i/bench_default.json
