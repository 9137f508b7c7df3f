#!/usr/bin/env python3
"""Experiment (GPU box): RNS products of config 5's shape with the limbs' launch chains on 1, 2 or 4 streams --
do the tails of the persistent XCD-local launches overlap with the next limb's launches?
usage: python3 tools/limb_streams_probe.py [--logn 17] [--batch 512] [--limbs 4] [--steps 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=17)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--limbs", type=int, default=4)
ap.add_argument("--bits", type=int, default=50)
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
n = 1 << a.logn
qs = [lib.find_prime(a.bits, n, k) for k in range(a.limbs)]
plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
per = a.batch * n
bufs = [lib.DeviceBuffer(a.limbs * per) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(4)]
def fill():
    for i, b in enumerate(bufs[:2]):
        for l, q in enumerate(qs):
            lib.fill_uniform(b.ptr + 8 * l * per, per, q, 1000 + i, l * per)
    lib.stream_sync()
def run(ns):
    if ns == 0:
        lib.rns_negacyclic_mul(plans, bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, a.batch)
    else:
        for l, p in enumerate(plans):
            off = 8 * l * per
            s = streams[l % ns].cuda_stream if ns > 1 else None
            p.negacyclic_mul(bufs[2].ptr + off, bufs[0].ptr + off, bufs[1].ptr + off, a.batch, stream=s)
    torch.cuda.synchronize()
sums = {}
for ns in (0, 1, 2, 4):
    fill(); run(ns)
    out = lib.DeviceBuffer(a.limbs * a.batch)
    lib.poly_checksum(out.ptr, bufs[2].ptr, n, a.limbs * a.batch); lib.stream_sync()
    sums[ns] = out.download().sum()
    ts = []
    for _ in range(a.steps):
        fill()
        t0 = time.perf_counter(); run(ns); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("N=2^%d limbs=%d batch=%d streams=%d: min %.3f median %.3f ms/step  %.0f RNS products/s (median)  checksum %s"
          % (a.logn, a.limbs, a.batch, ns, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, a.batch / ts[len(ts) // 2],
             "same" if sums[ns] == sums[0] else "DIFFERENT"))
