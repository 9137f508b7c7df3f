#!/bin/bash
# tools/r06_last_call.sh : the round's last gpurun command -- GPU suite, smoke(), then tools/r06_collect_and_rebench.sh (collection on the shipped
# library, bench lines quoting its own counters, a soak with a fresh seed)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06k_tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r06k_tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/r06_collect_and_rebench.sh r06k 300
