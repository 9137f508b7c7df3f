#!/usr/bin/env python3
"""tools/latency_probe.py: device time of ONE polynomial (batch 1, resident), block kernel against the column-pass-only engine (set_generic),
the policies the reference-signature entry points use"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
q = 0x7fffffffe0001
for lg in (10, 11, 12, 13, 14):
    n = 1 << lg
    w = lib.min_root(q, n)
    row = []
    for name, ar in (("u64", lib.ARITH_U64), ("r4", lib.ARITH_U64_R4), ("f64", lib.ARITH_F64)):
        for generic in (0, 1):
            if generic and ar == lib.ARITH_U64_R4: continue
            plan = lib.Plan(n, q, w, arith=ar)
            if generic: plan.set_generic(True)
            buf = lib.DeviceBuffer(2 * n)
            lib.fill_uniform(buf.ptr, 2 * n, q, 1)
            for batch in (1, 2):
                for _ in range(20): plan.fwd(buf.ptr, batch, lazy=True, wide=True)
                e0, e1 = lib.Event(), lib.Event()
                lib.stream_sync(); e0.record()
                for _ in range(200): plan.fwd(buf.ptr, batch, lazy=True, wide=True)
                e1.record(); us = e1.elapsed_ms_since(e0) * 1000 / 200
                row.append("%s%s b%d %.1f" % (name, " generic" if generic else "", batch, us))
            buf.free(); plan.destroy()
    print("2^%d: " % lg + " | ".join(row))
