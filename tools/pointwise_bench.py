#!/usr/bin/env python3
"""tools/pointwise_bench.py [BYTES]: ntt_pointwise_mul_batch (24 N bytes per product) on the library's grid (about four grid-stride
iterations per workgroup) against the 8192 workgroups of rounds 1-4 (NTT_OPT_MAX_GRID 8192), canonical and lazy words, three moduli;
the copy probes beside them.  (The sweep over grids that chose the rule: profiles/r05/pointwise_grid.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
n = 1 << 14
words = int(float(sys.argv[1]) if len(sys.argv) > 1 else 4e9) // 8
batch = words // n
bufs = [lib.DeviceBuffer(batch * n) for _ in range(3)]
def timed(fn, steps=10):
    for _ in range(3): fn()
    e0, e1 = lib.Event(), lib.Event()
    lib.stream_sync(); e0.record()
    for _ in range(steps): fn()
    e1.record(); return e1.elapsed_ms_since(e0) / steps
for rep in range(2):
    for q in (0x7fffffffe0001, 0xffffffff00001, 0xffffffffffc0001):
        plan = lib.Plan(n, q, lib.min_root(q, n))
        for i, b in enumerate(bufs): lib.fill_uniform(b.ptr, batch * n, q, 5 + i)
        for lazy in (False, True):
            row = []
            for grid in ((8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 22, 0) if 'sweep' in sys.argv else (8192, 0)):
                plan.set_option(lib.OPT_MAX_GRID, grid)
                ms = timed(lambda: plan.pointwise_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch, lazy_in=lazy))
                row.append("%s: %.3f" % (grid or "library grid (about four iterations per workgroup)", 24 * n * batch / ms / 1e6 / 8000))
            print("rep %d q %#x %s: %s" % (rep, q, "lazy" if lazy else "canonical", " | ".join(row)))
        plan.destroy()
ms = timed(lambda: lib.copy_probe(bufs[1].ptr, bufs[0].ptr, batch * n))
print("out-of-place copy probe: %.0f GB/s read + written (%.3f of 8 TB/s)" % (16 * batch * n / ms / 1e6, 16 * batch * n / ms / 1e6 / 8000))
ms = timed(lambda: lib.rmw_probe(bufs[0].ptr, batch * n))
print("in-place read-modify-write probe: %.0f GB/s (%.3f)" % (16 * batch * n / ms / 1e6, 16 * batch * n / ms / 1e6 / 8000))
