#!/usr/bin/env python3
"""tools/rns_chain_bench.py [--logn 14] [--batch 1]: RNS products over a modulus chain with primes of several sizes (a 60-bit first
prime, eight 50-bit primes, four 57-bit primes): the limb list served as runs of compatible limbs (one launch per pass and run)
against one launch chain per prime (NTT_RNS_LOOP=1)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, nargs="+", default=[14])
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
for ln in a.logn:
    n = 1 << ln
    qs = [lib.find_prime(60, n)] + [lib.find_prime(50, n, i) for i in range(8)] + [lib.find_prime(57, n, i) for i in range(4)]
    plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
    for batch in a.batch:
        per = batch * n
        bufs = [lib.DeviceBuffer(len(qs) * per) for _ in range(3)]
        def fill():
            for i, b in enumerate(bufs[:2]):
                for l, q in enumerate(qs):
                    lib.fill_uniform(b.ptr + 8 * l * per, per, q, 1000 + i, l * per)
        for loop in ("1", "0"):
            lib.set_rns_launch(plans, loop)
            fill(); lib.rns_negacyclic_mul(plans, bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch); lib.stream_sync()
            e0, e1 = lib.Event(), lib.Event()
            tot = 0.0
            for _ in range(a.steps):
                fill(); lib.stream_sync()
                e0.record(); lib.rns_negacyclic_mul(plans, bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch); e1.record()
                tot += e1.elapsed_ms_since(e0)
            print("N=2^%d 13 limbs (60 | 8 x 50 | 4 x 57 bits) batch=%d NTT_RNS_LOOP=%s: %.3f ms per RNS product step" % (ln, batch, loop, tot / a.steps))
        for b in bufs:
            b.free()
    for p in plans:
        p.destroy()
