#!/usr/bin/env python3
"""Experiment: a multi-pass transform (N = 2^16) issued from S host-side streams, each working through its own share
of the batch chunk by chunk, against one stream -- do the memory-bound column passes of one stream overlap the
FP64-bound block passes of another?  usage: python3 tools/exp_two_streams.py [--logn 16] [--streams 1 2 3 4]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=16)
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--bytes", type=float, default=8e9)
ap.add_argument("--chunk", type=int, nargs="+", default=[256])
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--op", default="fwd")
a = ap.parse_args()
n = 1 << a.logn
q = 0x80000001c0001
w = lib.min_root(q, n)
batch = int(a.bytes / (8 * n))
buf = lib.DeviceBuffer(batch * n)
lib.fill_uniform(buf.ptr, batch * n, q, 5)
bufs2 = None
if a.op == "mul":
    bufs2 = (lib.DeviceBuffer(batch * n), lib.DeviceBuffer(batch * n))
    lib.fill_uniform(bufs2[0].ptr, batch * n, q, 6)
for chunk in a.chunk:
    for S in a.streams:
        plans, streams = [], []
        for s in range(S):
            p = lib.Plan(n, q, w)
            p.set_option(lib.OPT_CHUNK_MIB, chunk)
            if a.grid:
                p.set_option(lib.OPT_MAX_GRID, a.grid)
            plans.append(p)
            h = C.c_void_p()
            lib._check(lib._lib.ntt_stream_create(0, C.byref(h)))
            streams.append(h.value)
        # interleave the shares chunk-wise: stream s takes polynomials [s*share, (s+1)*share)
        share = batch // S
        def run():
            for s in range(S):
                off = s * share * n * 8
                if a.op == "fwd":
                    plans[s].fwd(buf.ptr + off, share, stream=streams[s])
                elif a.op == "inv":
                    plans[s].inv(buf.ptr + off, share, stream=streams[s])
                else:
                    plans[s].negacyclic_mul(bufs2[1].ptr + off, buf.ptr + off, bufs2[0].ptr + off, share, stream=streams[s])
        def sync():
            for s in range(S):
                lib.stream_sync(0, streams[s])
        run(); sync()
        import time
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        sync()
        dt = (time.perf_counter() - t0) / a.steps
        per = 72 if a.op == "mul" else 16
        print("logn %d op %s chunk %4d MiB streams %d grid %d: %8.3f M/s  frac %.3f" % (a.logn, a.op, chunk, S, a.grid, S * share / dt / 1e6, S * share * per * n / dt / 8e12), flush=True)
        for s in range(S):
            lib._lib.ntt_stream_destroy(0, streams[s])
            plans[s].destroy()
