for r in 1 2 3; do
  for lib in build/libntt_prev.so optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so; do
    echo "== $lib"
    NTT_LIB=$lib timeout 100 python3 tools/sweep.py --logn 14 16 --ops fwd inv --arith f64 --bytes 8e9 --steps 10 2>&1 | tail -4
  done
done
timeout 100 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -2
