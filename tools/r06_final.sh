#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 tools/soak.py --seconds 420 --seed 6 > gpurun_out/r06c_soak_seed6.txt 2>&1; tail -3 gpurun_out/r06c_soak_seed6.txt
timeout 3300 bash tools/collect_r06.sh gpurun_out/r06c 2>&1 | tail -30
