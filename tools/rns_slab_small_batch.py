#!/usr/bin/env python3
"""tools/rns_slab_small_batch.py: the slab RNS entry points on SMALL batches (a few polynomials x many primes, [batch][limb][N]: one
ciphertext polynomial = its limbs side by side) -- one launch over a run of limbs (NTT_OPT_RNS_LAUNCH 0) against one launch chain per
limb (1) and the library's own choice; milliseconds per call, 200 calls back to back.  The companion of rns_pointer_small_batch.py:
does the automatic choice take the faster form at every size?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
print("# lib sha256 %s" % __import__("hashlib").sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()[:16])
print("%-5s %-6s %-6s %-34s %12s %12s %12s %8s" % ("logn", "limbs", "count", "call", "per limb ms", "one launch", "automatic", "auto/best"))
K = 3
for logn, nl, count in ((12, 16, 2), (13, 16, 2), (14, 16, 2), (14, 16, 8), (15, 16, 2), (16, 16, 2), (16, 16, 8), (17, 8, 4), (14, 4, 64), (16, 4, 64)):
    n = 1 << logn
    qs = [lib.find_prime(50, n, i) for i in range(nl)]
    plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
    lay = lib.batch_major(plans)
    words = nl * count * n
    bufs = [lib.DeviceBuffer(words) for _ in range(2 * K + 1)]
    for i, b in enumerate(bufs):
        lib.fill_uniform(b.ptr, words, min(qs), 1 + i, 0)
    key = lib.DeviceBuffer(K * nl * n)
    lib.fill_uniform(key.ptr, K * nl * n, min(qs), 30, 0)
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
    a, b, c = bufs[:K], bufs[K:2 * K], bufs[2 * K]
    keys = [key.ptr + 8 * nl * n * i for i in range(K)]
    rows = [
        ("forward", lambda: lib.rns_fwd(plans, a[0].ptr, count, layout=lay)),
        ("inverse", lambda: lib.rns_inv(plans, a[0].ptr, count, layout=lay)),
        ("c = a * b", lambda: lib.rns_negacyclic_mul(plans, c.ptr, a[0].ptr, b[0].ptr, count, layout=lay)),
        ("c = inv(sum_3 a_i^ . key_i^), shared", lambda: lib.rns_inv_dot(plans, c.ptr, [x.ptr for x in a], keys, count, lib.MUL_B_BROADCAST, layout=lay)),
        ("c^ += fwd(a) . key^, shared", lambda: lib.rns_fwd_mul(plans, c.ptr, a[0].ptr, keys[0], count, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST, layout=lay)),
    ]
    for name, fn in rows:
        ms = {}
        for mode in ("1", "0", None):
            lib.set_rns_launch(plans, mode)
            ms[mode] = timed(fn)
        print("%-5d %-6d %-6d %-34s %12.4f %12.4f %12.4f %8.2f" % (logn, nl, count, name, ms["1"], ms["0"], ms[None], ms[None] / min(ms["1"], ms["0"])))
    for x in bufs + [key]:
        x.free()
    for p in plans:
        p.destroy()
