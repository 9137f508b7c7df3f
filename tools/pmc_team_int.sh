#!/bin/bash
# tools/pmc_team_int.sh OUTDIR : fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, one rocprofv3 --pmc pass each) of the wide integer
# policy's one-launch transforms (team_kernel<ArithU64X<3>, LEAD, fwd>) at N = 2^15, 2^16, 2^17, 4 GiB slabs of a 57-bit modulus
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ln in 15 16 17; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/n$ln/$ctr -- python3 tools/sweep.py --logn $ln --ops fwd --qs 0x1fffffffffc0001 --bytes 4e9 --xcd-local 1 --steps 3 > $out/n${ln}_$ctr.log 2>&1
  done
  echo "## N = 2^$ln, 4 GiB slab (algorithmic 16 N per transform = 4096 MiB read + 4096 MiB written per launch); FETCH_SIZE / WRITE_SIZE in KiB, FETCH x 2 on gfx950" >> $out/pmc_team_integer.txt
  python3 tools/pmc_summary.py $out/n$ln team_kernel >> $out/pmc_team_integer.txt 2>&1
done
cat $out/pmc_team_integer.txt
