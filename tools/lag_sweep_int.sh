cd $GRAFT_REPO_ROOT
for lag in 0 3 5 6 8 10 12 16; do
  echo "== lag $lag"
  python3 tools/sweep.py --logn 15 16 17 --ops fwd --qs 0xffffffffffc0001 --bytes 4e9 --xcd-local 1 --lag $lag 2>&1 | grep -v "^logn"
done
for wpc in 2 3; do
  echo "== wpc $wpc"
  python3 tools/sweep.py --logn 15 16 17 --ops fwd --qs 0xffffffffffc0001 --bytes 4e9 --xcd-local 1 --wpc $wpc 2>&1 | grep -v "^logn"
done
