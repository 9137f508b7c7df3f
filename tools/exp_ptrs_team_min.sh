for mn in 64 32 16 8 2; do
  echo "== NTT_PTRS_TEAM_MIN=$mn"
  NTT_PTRS_TEAM_MIN=$mn timeout 200 python3 tools/rns_pointer_small_batch.py 2>&1 | grep -E "^(15|16|17) " | awk '$3 < 64'
done
