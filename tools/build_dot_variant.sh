#!/bin/bash
# tools/build_dot_variant.sh NAME "-DFLAG ..." : build/libntt_NAME.so = the working tree's library with the NTT-domain product
# kernels (inst_dot_*.hip) rebuilt with FLAGS (A/B builds of dot_inv_kernel's tuning knobs; the other objects are reused)
set -e
name=$1; flags=$2
csrc=optimized-number-theoretic-transform-implementations_amd/csrc
mkdir -p build/$name
pids=()
for f in inst_dot_f64k0 inst_dot_f64k1 inst_dot_f64k18 inst_dot_f64w inst_dot_u64; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags \
     -Iinclude -Iinclude/internal -I$csrc -c -o build/$name/$f.o $csrc/$f.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
others=$(ls $csrc/*.o | grep -v inst_dot_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libntt_$name.so build/$name/*.o $others
python3 tools/check_spills.py build/$name/inst_dot_f64k1.o | awk '$5>0' | grep -v "Lb1EEEv" | head -5
echo built build/libntt_$name.so
