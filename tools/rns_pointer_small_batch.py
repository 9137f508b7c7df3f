#!/usr/bin/env python3
"""tools/rns_pointer_small_batch.py: the RNS twins of the products over device pointer tables on SMALL batches -- a few separately
held ciphertext polynomials x many primes, the shape an FHE library hands over one operation at a time.  One launch over all limbs
of a run (the fused kernels' table-reading MULTI instances: NTT_OPT_RNS_LAUNCH 0) against one launch per limb (1: what the twins did
before), and the library's own choice; milliseconds per call, 200 calls back to back on one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ontt
lib = ontt.load()
print("# lib sha256 %s" % __import__("hashlib").sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()[:16])
print("%-5s %-6s %-6s %-34s %12s %12s %12s %8s" % ("logn", "limbs", "count", "product", "per limb ms", "one launch", "automatic", "speedup"))
K = 3
for logn, nl, count in ((14, 16, 1), (14, 16, 2), (14, 16, 4), (14, 4, 2), (13, 16, 2), (12, 16, 2), (12, 16, 8), (14, 16, 64), (14, 16, 256), (15, 16, 2), (16, 16, 2), (16, 8, 4), (17, 4, 2), (16, 16, 4), (16, 16, 8), (17, 8, 8), (16, 4, 64), (16, 4, 512), (15, 4, 512)):
    n = 1 << logn
    qs = [lib.find_prime(50, n, i) for i in range(nl)]
    plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
    rng = np.random.default_rng(logn + nl)
    nops = 2 * K + 1
    total = nops * count
    span = nl * n
    gaps = rng.integers(1, 2048, size=total) + np.arange(total) % 7
    starts = np.cumsum(gaps + span) - span
    words = int(starts[-1] + span + 8)
    order = rng.permutation(total)
    pool = lib.DeviceBuffer(words)
    lib.fill_uniform(pool.ptr, words, min(qs), 1, 0)
    tabs = [lib.DeviceBuffer(count).upload(np.array([pool.ptr + 8 * int(starts[i]) for i in order[o * count:(o + 1) * count]], dtype=np.uint64)) for o in range(nops)]
    key = lib.DeviceBuffer(K * nl * n)
    lib.fill_uniform(key.ptr, K * nl * n, min(qs), 3, 0)
    ev0, ev1 = lib.Event(0), lib.Event(0)

    def timed(fn, reps=200):
        for _ in range(5):
            fn()
        lib.stream_sync(0, None)
        ev0.record(None)
        for _ in range(reps):
            fn()
        ev1.record(None)
        return ev1.elapsed_ms_since(ev0) / reps
    ta, tb, tc = tabs[:K], tabs[K:2 * K], tabs[2 * K]
    keys = [key.ptr + 8 * nl * n * i for i in range(K)]
    rows = [
        ("c = a * b", lambda: lib.rns_negacyclic_mul_dev_ptrs(plans, tc.ptr, ta[0].ptr, tb[0].ptr, count, n)),
        ("c = inv(sum_3 a_i^ . key_i^), shared", lambda: lib.rns_inv_dot_dev_ptrs(plans, tc.ptr, [x.ptr for x in ta], keys, count, n, lib.MUL_B_BROADCAST)),
        ("c^ += fwd(a) . key^, shared", lambda: lib.rns_fwd_mul_dev_ptrs(plans, tc.ptr, ta[0].ptr, keys[0], count, n, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST)),
    ]
    for name, fn in rows:
        ms = {}
        for mode in ("1", "0", None):
            lib.set_rns_launch(plans, mode)
            ms[mode] = timed(fn)
        print("%-5d %-6d %-6d %-34s %12.4f %12.4f %12.4f %8.2f" % (logn, nl, count, name, ms["1"], ms["0"], ms[None], ms["1"] / ms["0"]))
    for x in tabs + [key, pool]:
        x.free()
    for p in plans:
        p.destroy()
