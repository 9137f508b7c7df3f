#!/usr/bin/env python3
"""One kernel family under a profiler (GPU box): N launches of ntt_inv_product_batch / ntt_inv_dot_batch / ntt_fwd_mul_batch and
nothing else on the device but the operand fills.  usage: python3 tools/dot_probe.py [--op dot|mul] [--logn 14] [--k 1] [--bcast]
[--acc] [--launches 6] [--bytes 4e9]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--op", choices=("dot", "mul"), default="dot")
ap.add_argument("--logn", type=int, default=14)
ap.add_argument("--k", type=int, default=1)
ap.add_argument("--bits", type=int, default=50)
ap.add_argument("--bcast", action="store_true")
ap.add_argument("--acc", action="store_true")
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--bytes", type=float, default=4e9)
a = ap.parse_args()
n = 1 << a.logn
q = lib.find_prime(a.bits, n)
plan = lib.Plan(n, q, lib.min_root(q, n))
batch = max(1, int(a.bytes / (8 * n)))
A = [lib.DeviceBuffer(batch * n) for _ in range(a.k)]
B = [lib.DeviceBuffer(batch * n) for _ in range(a.k)]
c = lib.DeviceBuffer(batch * n)
for i, x in enumerate(A + B + [c]):
    lib.fill_uniform(x.ptr, batch * n, q, 900 + i)
flags = (lib.MUL_B_BROADCAST if a.bcast else 0) | (lib.MUL_ACCUMULATE if (a.acc and a.op == "mul") else 0)
e0, e1 = lib.Event(), lib.Event()
lib.stream_sync()
e0.record()
for _ in range(a.launches):
    if a.op == "dot":
        plan.inv_dot(c.ptr, [x.ptr for x in A], [x.ptr for x in B], batch, flags)
    else:
        plan.fwd_mul(c.ptr, A[0].ptr, B[0].ptr, batch, flags)
e1.record()
ms = e1.elapsed_ms_since(e0) / a.launches
print("%s N=2^%d k=%d bcast=%d acc=%d batch=%d: %.3f ms per launch" % (a.op, a.logn, a.k, a.bcast, a.acc, batch, ms))
