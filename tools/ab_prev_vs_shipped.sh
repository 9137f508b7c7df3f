for i in 1 2 3; do
  for lib in build/libntt_prev.so ""; do
    if [ -n "$lib" ]; then export NTT_LIB=$lib; else unset NTT_LIB; fi
    python3 bench.py --steps 40 --warmup 9 --no-cpu-baseline --headline-only 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('${lib:-shipped}', d['lib_sha256'][:8], round(d['value']), round(r['frac'], 4), round(r.get('frac_at_min', 0), 4))"
  done
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | head -3
