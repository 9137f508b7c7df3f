#!/usr/bin/env python3
"""tools/chunk_sweep.py [--logn 16] [--q 0xffffffff00001] [--op inv]: chunk size (NTT_OPT_CHUNK_MIB) of the per-pass path of a
multi-pass transform, one plan, 4 GiB slab."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, nargs="+", default=[16])
ap.add_argument("--q", default="0xffffffff00001")
ap.add_argument("--op", default="inv")
ap.add_argument("--chunks", type=int, nargs="+", default=[64, 128, 192, 256, 384, 512])
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
q = int(a.q, 0)
for ln in a.logn:
    n = 1 << ln
    batch = (4 << 30) // (8 * n)
    plan = lib.Plan(n, q, lib.min_root(q, n))
    plan.set_option(lib.OPT_XCD_LOCAL, 0)
    buf = lib.DeviceBuffer(batch * n)
    lib.fill_uniform(buf.ptr, batch * n, q, 5)
    for mib in a.chunks:
        plan.set_option(lib.OPT_CHUNK_MIB, mib)
        run = (lambda: plan.inv(buf.ptr, batch)) if a.op == "inv" else (lambda: plan.fwd(buf.ptr, batch))
        for _ in range(2): run()
        e0, e1 = lib.Event(), lib.Event()
        lib.stream_sync(); e0.record()
        for _ in range(a.steps): run()
        e1.record(); ms = e1.elapsed_ms_since(e0) / a.steps
        print("2^%d %s q=%s chunk %4d MiB: %.3f ms  %.3f M NTT/s  frac %.3f" % (ln, a.op, a.q, mib, ms, batch / ms / 1e3, 16 * n * batch / ms / 1e6 / 8000))
    buf.free(); plan.destroy()
