#!/usr/bin/env python3
"""tools/pointer_product_bench.py: products over SEPARATELY HELD operands (every operand a device table of pointers to randomly
placed polynomials) against the same products over contiguous slabs.  Up to 2^14 the fused product kernels read their operands through
the tables (one launch per call, the slab forms' bytes), from 2^15 on the XCD-local one-launch kernels do; --chain switches those forms off (NTT_OPT_DOT_FUSED 0, NTT_OPT_FUSED_PRODUCT 0)
and measures what the entry points did before: table-reading element-wise kernels plus transforms over the tables.
Fractions are of 8 TB/s at the algorithmic bytes of the fused forms: 24N (c = a * b), (k + 1) 8N with a broadcast key / (2k + 1) 8N
without (c = inv(sum a_i^ . b_i^)), 24N / 32N (c^ (+)= fwd(a) . b^)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ontt
lib = ontt.load()
chain = "--chain" in sys.argv
Q = 0x7fffffffe0001
print("# lib sha256 %s%s" % (__import__("hashlib").sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()[:16], "  (--chain: fused table forms off)" if chain else ""))
print("%-5s %-6s %-44s %9s %9s %7s %7s" % ("logn", "count", "product", "slab ms", "tables ms", "frac", "vs slab"))
K = 3
for logn, count in ((14, 4096), (13, 8192), (12, 16384), (14, 512), (10, 65536), (15, 2048), (16, 1024), (17, 512)):   # (2^15..2^17: the XCD-local one-launch kernels)
    n = 1 << logn
    q = Q if (Q - 1) % (2 * n) == 0 else lib.find_prime(50, n, 0)
    plan = lib.Plan(n, q, lib.min_root(q, n))
    pt = lib.Plan(n, q, lib.min_root(q, n))      # the plan the table forms run on
    if chain:
        pt.set_option(lib.OPT_DOT_FUSED, 0)
        pt.set_option(lib.OPT_FUSED_PRODUCT, 0)
        pt.set_option(lib.OPT_XCD_LOCAL, 0)
    rng = np.random.default_rng(logn)
    nops = 2 * K + 1
    total = nops * count
    gaps = rng.integers(1, 2048, size=total) + np.arange(total) % 7
    starts = np.cumsum(gaps + n) - n
    words = int(starts[-1] + n + 8)
    order = rng.permutation(total)       # all operands' polynomials interleaved at random
    pool = lib.DeviceBuffer(words)
    lib.fill_uniform(pool.ptr, words, q, 1, 0)
    tabs = [lib.DeviceBuffer(count).upload(np.array([pool.ptr + 8 * int(starts[i]) for i in order[o * count:(o + 1) * count]], dtype=np.uint64)) for o in range(nops)]
    slabs = [lib.DeviceBuffer(count * n) for _ in range(nops)]
    for s_ in slabs:
        lib.fill_uniform(s_.ptr, count * n, q, 2, 0)
    key = lib.DeviceBuffer(K * n)
    lib.fill_uniform(key.ptr, K * n, q, 3, 0)
    ev0, ev1 = lib.Event(0), lib.Event(0)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        lib.stream_sync(0, None)
        ev0.record(None)
        for _ in range(reps):
            fn()
        ev1.record(None)
        return ev1.elapsed_ms_since(ev0) / reps
    a, b, c = slabs[:K], slabs[K:2 * K], slabs[2 * K]
    ta, tb, tc = tabs[:K], tabs[K:2 * K], tabs[2 * K]
    keys = [key.ptr + 8 * n * i for i in range(K)]
    rows = [
        ("c = a * b", 24,
         lambda: plan.negacyclic_mul(c.ptr, a[0].ptr, b[0].ptr, count), lambda: pt.negacyclic_mul_dev_ptrs(tc.ptr, ta[0].ptr, tb[0].ptr, count)),
        ("c = inv(a^ . b^)", 24,
         lambda: plan.inv_dot(c.ptr, [a[0].ptr], [b[0].ptr], count), lambda: pt.inv_dot_dev_ptrs(tc.ptr, [ta[0].ptr], [tb[0].ptr], count)),
        ("c = inv(sum_3 a_i^ . b_i^)", 8 * (2 * K + 1),
         lambda: plan.inv_dot(c.ptr, [x.ptr for x in a], [x.ptr for x in b], count), lambda: pt.inv_dot_dev_ptrs(tc.ptr, [x.ptr for x in ta], [x.ptr for x in tb], count)),
        ("c = inv(sum_3 a_i^ . key_i^), key shared", 8 * (K + 1),
         lambda: plan.inv_dot(c.ptr, [x.ptr for x in a], keys, count, lib.MUL_B_BROADCAST), lambda: pt.inv_dot_dev_ptrs(tc.ptr, [x.ptr for x in ta], keys, count, lib.MUL_B_BROADCAST)),
        ("c^ = fwd(a) . b^", 24,
         lambda: plan.fwd_mul(c.ptr, a[0].ptr, b[0].ptr, count), lambda: pt.fwd_mul_dev_ptrs(tc.ptr, ta[0].ptr, tb[0].ptr, count)),
        ("c^ += fwd(a) . key^, key shared", 24,
         lambda: plan.fwd_mul(c.ptr, a[0].ptr, keys[0], count, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST),
         lambda: pt.fwd_mul_dev_ptrs(tc.ptr, ta[0].ptr, keys[0], count, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST)),
    ]
    for name, bpn, slab_fn, tab_fn in rows:
        ms_s, ms_t = timed(slab_fn), timed(tab_fn)
        print("%-5d %-6d %-44s %9.3f %9.3f %7.3f %7.3f" % (logn, count, name, ms_s, ms_t, count * bpn * n / (ms_t * 1e-3) / 8e12, ms_s / ms_t))
    for x in tabs + slabs + [key, pool]:
        x.free()
    plan.destroy(), pt.destroy()
