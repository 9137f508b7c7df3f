#!/bin/bash
# tools/r06_collect_and_rebench.sh TAG [SOAK_SECONDS] : one gpurun command -- the evidence collection on the shipped library, its counter summaries
# installed into the box's copy of profiles/r06, then the bench LINES taken again so that each quotes the counters of the binary it ran
# on (`traffic_from_this_binary: true`; tools/rebench_after_install.sh), then a soak with a fresh seed.  Locally afterwards:
#   bash tools/install_collection.sh gpurun_out/TAG && cp gpurun_out/TAG_lines/*.json profiles/r06/
tag=$1; soak=${2:-0}
bash tools/collect_r06.sh gpurun_out/$tag > gpurun_out/collect_$tag.log 2>&1
bash tools/install_collection.sh gpurun_out/$tag > /dev/null
bash tools/rebench_after_install.sh gpurun_out/${tag}_lines $soak
cat gpurun_out/$tag/LIBRARY_SHA256; head -c 300 gpurun_out/${tag}_lines/bench_default.json
