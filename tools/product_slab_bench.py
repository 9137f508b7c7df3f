#!/usr/bin/env python3
"""tools/product_slab_bench.py: c = a * b over contiguous slabs, N = 2^8 .. 2^14 (the fused product kernels), 50- and 52-bit moduli; one line
per size: ms per call and the fraction of 8 TB/s at 24N bytes.  NTT_LIB selects an A/B library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
print("# lib sha256 %s" % __import__("hashlib").sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()[:16])
for bits in (50, 52):
    for logn in (14, 13, 12, 10):
        n = 1 << logn
        count = (1 << 26) >> logn
        q = lib.find_prime(bits, n, 0)
        plan = lib.Plan(n, q, lib.min_root(q, n))
        a, b, c = (lib.DeviceBuffer(count * n) for _ in range(3))
        for i, x in enumerate((a, b)):
            lib.fill_uniform(x.ptr, count * n, q, 5 + i, 0)
        ev0, ev1 = lib.Event(0), lib.Event(0)
        for _ in range(5):
            plan.negacyclic_mul(c.ptr, a.ptr, b.ptr, count)
        lib.stream_sync(0, None)
        best = []
        for _ in range(3):
            ev0.record(None)
            for _ in range(20):
                plan.negacyclic_mul(c.ptr, a.ptr, b.ptr, count)
            ev1.record(None)
            best.append(ev1.elapsed_ms_since(ev0) / 20)
        ms = sorted(best)[1]
        print("bits %d logn %2d count %6d  %.3f ms  frac %.3f" % (bits, logn, count, ms, count * 24 * n / (ms * 1e-3) / 8e12))
        a.free(), b.free(), c.free(), plan.destroy()
