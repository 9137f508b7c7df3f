#!/usr/bin/env python3
"""tools/pointer_batch_bench.py: randomly placed polynomials against one contiguous slab (review r05 item 2: "4096 randomly placed
2^14-polynomials within 3 % of the contiguous slab").  Per size: the slab through ntt_fwd_batch, the same polynomials scattered over a
larger buffer (random gaps, shuffled order) through ntt_transform_dev_ptrs (device table, as given), ntt_transform_ptrs (host array:
sort + overlap check + upload per call), and -- what rounds 1-5 did with such a batch -- one launch per polynomial."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ontt
lib = ontt.load()
Q = 0x7fffffffe0001
print("# lib sha256 %s" % __import__("hashlib").sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()[:16])
print("%-5s %-6s %-34s %9s %10s %7s %9s" % ("logn", "count", "form", "ms/call", "M NTT/s", "frac", "vs slab"))
for logn, count in ((14, 4096), (14, 32768), (12, 16384), (13, 8192), (16, 1024), (17, 512), (10, 65536)):
    n = 1 << logn
    q = Q if (Q - 1) % (2 * n) == 0 else lib.find_prime(50, n, 0)
    plan = lib.Plan(n, q, lib.min_root(q, n))
    rng = np.random.default_rng(logn)
    gaps = rng.integers(1, 4096, size=count) + np.arange(count) % 7
    starts = np.cumsum(gaps + n) - n
    words = int(starts[-1] + n + 8)
    order = rng.permutation(count)
    d = lib.DeviceBuffer(words)
    lib.fill_uniform(d.ptr, words, q, 1, 0)
    ptrs = [d.ptr + 8 * int(starts[i]) for i in order]
    tab = lib.DeviceBuffer(count).upload(np.array(ptrs, dtype=np.uint64))
    stab = lib.DeviceBuffer(count).upload(np.array(sorted(ptrs), dtype=np.uint64))
    ev0, ev1 = lib.Event(0), lib.Event(0)
    import ctypes as C
    harr = (C.c_void_p * count)(*ptrs)     # the caller's host array, built once (a C caller has it lying around)

    def timed(fn, reps):
        for _ in range(3):
            fn()
        lib.stream_sync(0, None)
        ev0.record(None)
        for _ in range(reps):
            fn()
        ev1.record(None)
        return ev1.elapsed_ms_since(ev0) / reps
    reps = 20
    rows = [("contiguous slab, ntt_fwd_batch", timed(lambda: plan.fwd(d.ptr, count), reps)),
            ("scattered, device table (shuffled)", timed(lambda: plan.transform_dev_ptrs(tab.ptr, count), reps)),
            ("scattered, device table (sorted)", timed(lambda: plan.transform_dev_ptrs(stab.ptr, count), reps)),
            ("scattered, host array (per call: sort, check, upload)", timed(lambda: lib._check(lib._lib.ntt_transform_ptrs(plan.h, harr, count, 0, None)), reps))]
    if count <= 4096:
        def one_by_one():
            for p in ptrs:
                plan.fwd(p, 1)
        rows.append(("scattered, one launch per polynomial (r05)", timed(one_by_one, 2)))
    for name, ms in rows:
        print("%-5d %-6d %-34s %9.3f %10.2f %7.3f %9.3f" % (logn, count, name[:54], ms, count / ms / 1e3, count * 16 * n / (ms * 1e-3) / 8e12, rows[0][1] / ms))
    tab.free(), stab.free(), d.free(), plan.destroy()
