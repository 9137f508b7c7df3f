#!/usr/bin/env python3
"""Products of operands in the NTT domain (GPU box): c = inv(sum_{i<k} a_i^ (.) b_i^) and c = inv(fwd(a) (.) b^), the
one-launch kernels against the launches they replace (pointwise product(s) + inverse transform; forward + pointwise +
inverse), same buffers, same box, HIP events.
usage: python3 tools/domain_bench.py [--logn 12 14 16 17] [--k 1 2 3 8] [--bytes 4e9] [--bits 50] [--steps 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, nargs="+", default=[12, 14, 16, 17])
ap.add_argument("--k", type=int, nargs="+", default=[1, 2, 3, 8])
ap.add_argument("--bytes", type=float, default=4e9, help="bytes of ONE operand")
ap.add_argument("--bits", type=int, nargs="+", default=[50])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--arith", default="auto")
ap.add_argument("--no-broadcast", action="store_true")
ap.add_argument("--chunk-mib", type=int, default=0)
ap.add_argument("--xcd-local", type=int, default=-1, help="N = 2^15..2^17: 1 = both passes as items of one launch (team_dot_kernel), 0 = per-chunk launches, -1 = the library's choice")
ap.add_argument("--lag", type=int, default=0, help="--xcd-local 1: polynomials between the two passes (0 = default)")
ap.add_argument("--max-grid", type=int, default=0, help="cap on workgroups per launch (0 = the kernels' own choice)")
ap.add_argument("--oversub", type=int, default=0, help="persistent block kernels: workgroups per resident slot (0 = the library's choice)")
a = ap.parse_args()
AR = {"auto": lib.ARITH_AUTO, "u64": lib.ARITH_U64, "f64": lib.ARITH_F64}


def timed(fn, steps):
    for _ in range(2):
        fn()
    e0, e1 = lib.Event(), lib.Event()
    lib.stream_sync()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    return e1.elapsed_ms_since(e0) / steps


print("%-5s %-5s %-3s %-9s %8s %10s %10s %8s %14s %7s" % ("logn", "bits", "k", "operands", "batch", "fused ms", "unfused ms", "speedup",
                                                          "M outputs/s", "frac"))
for bits in a.bits:
    for ln in a.logn:
        n = 1 << ln
        q = lib.find_prime(bits, n)
        plan = lib.Plan(n, q, lib.min_root(q, n), arith=AR[a.arith])
        if a.chunk_mib:
            plan.set_option(lib.OPT_CHUNK_MIB, a.chunk_mib)
        if a.oversub:
            plan.set_option(lib.OPT_BLOCK_OVERSUB, a.oversub)
        if a.max_grid:
            plan.set_option(lib.OPT_MAX_GRID, a.max_grid)
        plan.set_option(lib.OPT_XCD_LOCAL, a.xcd_local)
        plan.set_option(lib.OPT_XCD_LOCAL_LAG, a.lag)
        if os.environ.get("NTT_INT_WIDE") == "0" and plan.info()["arith"] == lib.ARITH_U64:
            plan.set_option(lib.OPT_INT_WIDE, 0)   # (tools/ab_int_wide_domain.sh)
        kmax = max(a.k)
        batch = max(1, int(a.bytes / (8 * n)))
        abufs = [lib.DeviceBuffer(batch * n) for _ in range(kmax)]
        bbufs = [lib.DeviceBuffer(batch * n) for _ in range(kmax)]
        c = lib.DeviceBuffer(batch * n)
        for i, x in enumerate(abufs + bbufs):
            lib.fill_uniform(x.ptr, batch * n, q, 300 + i)
        for k in a.k:
            for bcast in ((False,) if a.no_broadcast else (False, True)):
                flags = lib.MUL_B_BROADCAST if bcast else 0
                ap_, bp_ = [x.ptr for x in abufs[:k]], [x.ptr for x in bbufs[:k]]
                run = lambda: plan.inv_dot(c.ptr, ap_, bp_, batch, flags)
                plan.set_option(lib.OPT_DOT_FUSED, 1)
                ms = timed(run, a.steps)
                plan.set_option(lib.OPT_DOT_FUSED, 0)            # k pointwise(-accumulate) launches + the inverse transform
                ms0 = timed(run, max(2, a.steps // 2))
                plan.set_option(lib.OPT_DOT_FUSED, 1)
                byts = ((8 if bcast else 16) * k + 8) * n * batch   # operands in, c out (a broadcast key comes from the L2)
                print("%-5d %-5d %-3d %-9s %8d %10.3f %10.3f %8.2f %14.3f %7.3f"
                      % (ln, bits, k, "bcast b^" if bcast else "a^, b^", batch, ms, ms0, ms0 / ms, batch / ms / 1e3, byts / ms / 1e6 / 8000))
        # c = inv(fwd(a) (.) b^): one launch (three-pass form above 2^14) against fwd (lazy), pointwise, inverse
        def unfused():
            plan.fwd(abufs[0].ptr, batch, lazy=True)
            plan.pointwise_mul(c.ptr, abufs[0].ptr, bbufs[0].ptr, batch, lazy_in=True)
            plan.inv(c.ptr, batch)
        ms = timed(lambda: plan.mul_transformed(c.ptr, abufs[0].ptr, bbufs[0].ptr, batch), a.steps)
        ms0 = timed(unfused, max(2, a.steps // 2))
        print("%-5d %-5d %-3s %-9s %8d %10.3f %10.3f %8.2f %14.3f %7.3f"
              % (ln, bits, "-", "fwd(a),b^", batch, ms, ms0, ms0 / ms, batch / ms / 1e3, 24 * n * batch / ms / 1e6 / 8000))
        # c^ = fwd(a) (.) b^ and c^ += fwd(a) (.) b^ (result in the NTT domain): one launch against forward + pointwise
        for bcast, acc in ((False, False), (True, False), (False, True), (True, True)):
            flags = (lib.MUL_B_BROADCAST if bcast else 0) | (lib.MUL_ACCUMULATE if acc else 0)
            run = lambda: plan.fwd_mul(c.ptr, abufs[0].ptr, bbufs[0].ptr, batch, flags)
            plan.set_option(lib.OPT_DOT_FUSED, 1)
            ms = timed(run, a.steps)
            plan.set_option(lib.OPT_DOT_FUSED, 0)                # forward transform, then a pointwise (accumulate) launch
            ms0 = timed(run, max(2, a.steps // 2))
            plan.set_option(lib.OPT_DOT_FUSED, 1)
            byts = (16 + (0 if bcast else 8) + (8 if acc else 0)) * n * batch
            print("%-5d %-5d %-3s %-9s %8d %10.3f %10.3f %8.2f %14.3f %7.3f"
                  % (ln, bits, "mac" if acc else "mul", "fwd(a)" + (".key" if bcast else ".b^"), batch, ms, ms0, ms0 / ms, batch / ms / 1e3,
                     byts / ms / 1e6 / 8000))
        for x in abufs + bbufs + [c]:
            x.free()
        plan.destroy()
