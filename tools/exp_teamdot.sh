#!/bin/bash
# tools/exp_teamdot.sh OUTDIR : the NTT-domain products at N = 2^15..2^17 as ONE launch (team_dot_kernel, --xcd-local 1) against the
# per-chunk launches (--xcd-local 0), alternating, k = 1 and 3; lag sweep; then the GPU tests and the 52-bit inverse once more
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "xcd_local_ntt_domain or inv_dot or rns_products" > $out/pytest_dot.txt 2>&1; tail -3 $out/pytest_dot.txt
(for rep in 1 2; do for x in 0 1; do echo "== rep $rep xcd-local $x"; timeout 600 python3 tools/domain_bench.py --logn 15 16 17 --k 1 3 --steps 6 --xcd-local $x --no-broadcast | grep -v "^logn" | awk '{printf "2^%s k=%s %s: %s ms frac %s | ", $1, $3, $4, $(NF-4), $NF} END {print ""}'; done; done) > $out/domain_xcd_local.txt 2>&1
cat $out/domain_xcd_local.txt
(for lag in 4 6 8 10 14 20; do echo "== lag $lag: $(timeout 600 python3 tools/domain_bench.py --logn 15 16 17 --k 1 3 --steps 6 --xcd-local 1 --lag $lag --no-broadcast | grep "a^, b^" | awk '{printf "2^%s k=%s %s | ", $1, $3, $NF}')"; done) > $out/domain_xcd_local_lag.txt 2>&1
cat $out/domain_xcd_local_lag.txt
(for x in 0 1; do echo "== 60-bit xcd-local $x: $(timeout 600 python3 tools/domain_bench.py --logn 15 16 17 --k 1 3 --steps 6 --bits 60 --xcd-local $x --no-broadcast | grep "a^, b^" | awk '{printf "2^%s k=%s %s | ", $1, $3, $NF}')"; done; for x in 0 1; do echo "== broadcast key xcd-local $x: $(timeout 600 python3 tools/domain_bench.py --logn 16 17 --k 1 3 --steps 6 --xcd-local $x | grep "bcast" | awk '{printf "2^%s k=%s %s | ", $1, $3, $NF}')"; done) > $out/domain_xcd_local_other.txt 2>&1
cat $out/domain_xcd_local_other.txt
S="python3 tools/sweep.py --bytes 8e9 --steps 10 --qs 0xffffffff00001"
(for rep in 1 2; do for lib in build/libntt_prev.so ""; do
    echo "rep $rep ${lib:-this build}: $(NTT_LIB=$lib timeout 300 $S --logn 12 14 16 --ops inv mul --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
done; done) > $out/ab_52bit_inverse2.txt 2>&1
cat $out/ab_52bit_inverse2.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
for r in 1 2; do timeout 600 python3 bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('config 3: value %.4g frac %.3f kernel_ms %.3f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; done
