#!/bin/bash
# tools/install_collection.sh SRC : copies what tools/collect_r06.sh left under SRC (gpurun_out/...) into profiles/r06 -- the top-level files
# of the collection (bench lines, stamped kernel-stats CSVs, per-kernel tables, sweeps, soak) and the pmc_traffic*.json files bench.py quotes;
# the raw rocprofv3 directories stay in gpurun_out (scratch).  tests/test_abi.py then ties the installed evidence to the built library.
src=$1; dst=profiles/r06
for f in $src/*; do
  [ -f "$f" ] || continue
  case $(basename $f) in *.log|*.err|pmc_traffic_json.log) continue;; esac
  cp $f $dst/
done
cp $src/json/pmc_traffic*.json $dst/
cat $dst/LIBRARY_SHA256
