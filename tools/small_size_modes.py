#!/usr/bin/env python3
"""tools/small_size_modes.py [--logn 9]: the two throughput modes of the small transforms (profiles/r04/NOTES.txt): one process, one
allocation, the same transform at several byte offsets into it and repeated -- does the mode follow the buffer's placement?"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=9)
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--max-grid", type=int, nargs="+", default=[0], help="workgroup caps to try (0 = the library's grid)")
ap.add_argument("--allocs", type=int, default=2, help="allocations of --gib each to visit")
ap.add_argument("--ops", nargs="+", default=["fwd"], choices=["fwd", "inv"])
ap.add_argument("--no-offsets", action="store_true", help="skip the byte-offset walk over the first allocation")
a = ap.parse_args()
n = 1 << a.logn
q = 0x80000001c0001
plan = lib.Plan(n, q, lib.min_root(q, n))
words = int(a.gib * 2**30 / 8)
batch = words // n
pad = (64 << 20) // 8
buf = lib.DeviceBuffer(words + pad)
lib.fill_uniform(buf.ptr, words + pad, q, 3)
def rate(ptr, grid=0, op="fwd"):
    plan.set_option(lib.OPT_MAX_GRID, grid)
    run = plan.fwd if op == "fwd" else plan.inv
    for _ in range(4): run(ptr, batch)
    e0, e1 = lib.Event(), lib.Event()
    lib.stream_sync(); e0.record()
    for _ in range(10): run(ptr, batch)
    e1.record(); ms = e1.elapsed_ms_since(e0) / 10
    return 16 * n * batch / ms / 1e6 / 8000
print("pid %d base %#x (mod 2 MiB: %#x)" % (os.getpid(), buf.ptr, buf.ptr % (2 << 20)))
for rep in range(0 if a.no_offsets else 2):
    for off in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 32 << 20, 48 << 20):
        print("  rep %d offset %9d B: frac %.3f" % (rep, off, rate(buf.ptr + off)))
bufs = [buf] + [lib.DeviceBuffer(words) for _ in range(a.allocs - 1)]
for b in bufs[1:]: lib.fill_uniform(b.ptr, words, q, 4)
print("library %s" % os.environ.get("NTT_LIB", "(tree)"))
for op in a.ops:
    for g in a.max_grid:
        print("%s grid cap %6d: " % (op, g) + "  ".join("alloc %d (base %#x) %.3f" % (i, b.ptr, rate(b.ptr, g, op)) for i, b in enumerate(bufs))
              + "  | again: " + " ".join("%.3f" % rate(b.ptr, g, op) for b in bufs))
