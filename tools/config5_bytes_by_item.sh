#!/bin/bash
# tools/config5_bytes_by_item.sh OUTDIR : FETCH_SIZE / WRITE_SIZE of config 5's one launch per item type (diagnostic builds build/libntt_only{1,8,2,4}.so,
# tools/build_tu_variant.sh onlyV inst_team_f64k1 -DNTT_TEAMPROD_ONLY=V) and for the shipped library; summary: tools/config5_bytes_by_item.py
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export NTT_BENCH_NOCHECK=1       # (the diagnostic builds compute garbage on purpose)
for v in 1 8 2 4 15; do
  if [ $v = 15 ]; then unset NTT_LIB; else export NTT_LIB=build/libntt_only$v.so; [ -f $NTT_LIB ] || continue; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/only$v/$ctr -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/only${v}_$ctr.log 2>&1
  done
done
unset NTT_LIB
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $ctr | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/only15/$n -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/only15_$n.log 2>&1
done
unset NTT_BENCH_NOCHECK
python3 tools/config5_bytes_by_item.py $out
