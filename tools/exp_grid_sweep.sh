for rep in 1 2; do
for g in 0 256 248 240 232 224 208 192 160; do
  echo "== max_grid $g"
  timeout 120 python3 tools/sweep.py --logn 14 --ops fwd --qs 0x7fffffffe0001 --bytes 16e9 --steps 12 --max-grid $g 2>&1 | tail -1
done
done
