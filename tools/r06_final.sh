#!/bin/bash
# scratch: the evidence collection on the shipped library
bash tools/collect_r06.sh gpurun_out/r06e > gpurun_out/collect_r06e.log 2>&1
tail -3 gpurun_out/collect_r06e.log; cat gpurun_out/r06e/LIBRARY_SHA256; head -c 600 gpurun_out/r06e/bench_default.json
