#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : builds build/libntt_NAME.so (A/B and ablation builds)
set -e
name=$1; flags=$2
mkdir -p build/$name
pids=()
for f in $(cd optimized-number-theoretic-transform-implementations_amd/csrc && ls ntt_host.hip inst_*.hip | sed 's/\.hip$//'); do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags \
     -Iinclude -Iinclude/internal -Ioptimized-number-theoretic-transform-implementations_amd/csrc \
     -c -o build/$name/$f.o optimized-number-theoretic-transform-implementations_amd/csrc/$f.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libntt_$name.so build/$name/*.o
echo built build/libntt_$name.so
