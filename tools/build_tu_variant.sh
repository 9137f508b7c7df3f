#!/bin/bash
# tools/build_tu_variant.sh NAME PREFIX "-DFLAG ..." : build/libntt_NAME.so = the working tree's library with the translation units
# csrc/PREFIX*.hip (e.g. inst_dot_, inst_mul_) rebuilt with FLAGS; the other objects are reused (A/B builds of one kernel family)
set -e
name=$1; prefix=$2; flags=$3
csrc=optimized-number-theoretic-transform-implementations_amd/csrc
mkdir -p build/$name
pids=()
for f in $(cd $csrc && ls ${prefix}*.hip | sed 's/\.hip$//'); do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags \
     -Iinclude -Iinclude/internal -I$csrc -c -o build/$name/$f.o $csrc/$f.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
others=$(ls $csrc/*.o | grep -v "/${prefix}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libntt_$name.so build/$name/*.o $others
python3 tools/check_spills.py build/$name/*.o | awk '$5>0' | grep -v "Lb1EEEv" | head -5
echo built build/libntt_$name.so
