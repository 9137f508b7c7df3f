#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the persistent forward kernel (needs a -DNTT_STAMPS build:
tools/build_variant.sh stamps "-DNTT_STAMPS";  NTT_LIB=build/libntt_stamps.so python3 tools/stamps.py)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 14   # 12 or 14: the persistent forward loops carry the stamps
N, Q = 1 << LOGN, 0x7fffffffe0001
W = lib.min_root(Q, N)
batch = (1 << 31) // N                                   # 16 GiB
GRID = 256 * (4 if LOGN == 12 else 1)                    # resident workgroups (launch_fused)
plan = lib.Plan(N, Q, W)
buf = lib.DeviceBuffer(batch * N)
lib.fill_uniform(buf.ptr, batch * N, Q, 1)
plan.fwd(buf.ptr, batch); lib.stream_sync()
f = lib._lib.ntt_debug_stamps
f.argtypes = [C.c_void_p, C.c_int]
assert f(None, 1) == 0
e0, e1 = lib.Event(), lib.Event()
e0.record(); plan.fwd(buf.ptr, batch); e1.record()
launch_ms = e1.elapsed_ms_since(e0)
out = np.zeros((256, 16, 12), dtype=np.uint64)
assert f(out.ctypes.data, 0) == 0
iters = batch / GRID
names = ["wait prefetch+convert", "group A", "exch A->B (2 barriers)", "prefetch issue + group B", "exch B->C", "group C",
         "exch C->D", "group D", "-", "-", "final reduce + stores", "-"]
nw = N // 16 // 64                                        # waves of a workgroup
out = out[:, :nw, :]
per = out.astype(np.float64).mean(axis=(0, 1)) / iters
tot = per.sum()
# kernel 0 of the symbol lives in ONE instantiation TU; only the F64 K0 kernel ran
print("N = 2^%d: " % LOGN, end=""); print("cycles per block-iteration (mean over 256 WGs x their waves): total %.0f (s_memtime ticks = shader cycles)" % tot)
for n, v in zip(names, per):
    if v > 0: print("  %-28s %8.0f  %5.1f%%" % (n, v, 100 * v / tot))
spread = out.astype(np.float64).sum(axis=2).mean(axis=0) / iters
print("per-wave totals min/max: %.0f / %.0f" % (spread.min(), spread.max()))
# in-kernel shader clock under this load: cycles a wave counts per block iteration / wall time of an iteration (one launch =
# batch / 256 iterations per workgroup; the stamp build itself is a few per cent slower than the shipped kernel)
print("launch %.3f ms (stamp build) = %.2f us per block iteration -> in-kernel shader clock %.2f GHz" % (launch_ms, launch_ms * 1e3 / iters, tot / (launch_ms * 1e3 / iters) / 1e3))
