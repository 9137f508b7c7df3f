#!/bin/bash
# tools/folded_8_shards.sh OUT : the driver's 8-GPU run rehearsed on ONE GPU (review r05, item 1c/1d).
#   * `bench.py --gpus 8` in one process with the eight shards folded onto device 0 (NTT_BENCH_DEVICE_MOD=1): eight streams, eight
#     event chains, one host clock -- against ONE shard holding the same total batch.  Config 4 (8 x 131072 x 2^14 = 128 GiB on one
#     device: also the `--scaling strong --gpus 1` line, "no cliff at 128 GiB") and config 2 (0.8 ms steps, 16 host calls per step).
#   * the same eight shards as eight ranks under torch.distributed.run (gloo control plane).
out=$1; mkdir -p $(dirname $out)
cd $GRAFT_REPO_ROOT
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
r=d['roofline']
print('%-46s n_gpus=%d batch/gpu=%-7d value=%.4g %s  ms/step=%.3f  frac(slowest shard)=%.3f  aggregate frac=%.3f  parity=%s/%s cpu_baseline=%s' % (
  '$2', d['n_gpus'], d['config']['batch_per_gpu'], d['value'], d['unit'], d['ms_per_step'], r['frac'],
  d['value']*r['algorithmic_bytes_per_unit']/1e9/8000.0, d['parity']['shards_checked'], d['parity']['of'],
  ('%.4g %s on %d threads' % (d['cpu_baseline']['value'], d['cpu_baseline']['unit'], d['cpu_baseline']['threads'])) if 'cpu_baseline' in d else 'absent'))
"; }
{
echo "# folded 8-shard rehearsal on one MI355X, $(date -u +%Y-%m-%dT%H:%MZ), lib sha256 $(sha256sum optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so | cut -c1-16)"
echo "# 'aggregate frac' = whole-job units/s x algorithmic bytes / 8 TB/s: comparable between one shard and eight folded shards"
for rep in 1 2; do
  timeout 900 python3 bench.py --gpus 1 --scaling strong --steps 10 --warmup 4 --no-also --cpu-budget-s 3 > /tmp/f_one4.json 2>/tmp/f_one4.err || tail -3 /tmp/f_one4.err
  line /tmp/f_one4.json "config 4, ONE shard of 2^20 (strong, 128 GiB)"
  NTT_BENCH_DEVICE_MOD=1 timeout 900 python3 bench.py --gpus 8 --steps 10 --warmup 4 --no-also --cpu-budget-s 3 > /tmp/f_eight4.json 2>/tmp/f_eight4.err || tail -3 /tmp/f_eight4.err
  line /tmp/f_eight4.json "config 4, EIGHT folded shards of 131072"
  timeout 900 python3 bench.py --gpus 1 --config 2 --batch 524288 --steps 100 --warmup 80 --no-also --cpu-budget-s 3 > /tmp/f_one2.json 2>/tmp/f_one2.err || tail -3 /tmp/f_one2.err
  line /tmp/f_one2.json "config 2, ONE shard of 524288"
  NTT_BENCH_DEVICE_MOD=1 timeout 900 python3 bench.py --gpus 8 --config 2 --steps 100 --warmup 80 --no-also --cpu-budget-s 3 > /tmp/f_eight2.json 2>/tmp/f_eight2.err || tail -3 /tmp/f_eight2.err
  line /tmp/f_eight2.json "config 2, EIGHT folded shards of 65536"
done
NTT_BENCH_DEVICE_MOD=1 NTT_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29551 \
   bench.py --gpus 8 --steps 10 --warmup 4 --no-also --cpu-budget-s 3 > /tmp/f_ranks4.json 2>/tmp/f_ranks4.err || tail -5 /tmp/f_ranks4.err
line /tmp/f_ranks4.json "config 4, EIGHT RANKS (torch.distributed.run, gloo) folded"
NTT_BENCH_DEVICE_MOD=1 NTT_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29552 \
   bench.py --gpus 8 --config 2 --steps 100 --warmup 80 --no-also --cpu-budget-s 3 > /tmp/f_ranks2.json 2>/tmp/f_ranks2.err || tail -5 /tmp/f_ranks2.err
line /tmp/f_ranks2.json "config 2, EIGHT RANKS folded"
} > $out 2>&1
cp /tmp/f_one4.json $(dirname $out)/bench_config4_strong_one_gpu_128GiB.json
cp /tmp/f_eight4.json $(dirname $out)/bench_config4_eight_folded_shards.json
cat $out
